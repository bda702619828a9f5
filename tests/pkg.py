"""Imports the product package (its directory name has a hyphen, so importlib is needed)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
P = importlib.import_module("gmerlin-avdecoder_amd")
