import sys, os
sys.path.insert(0,'tests')
os.environ['MI_RTJ_SPEC']='1'
import numpy as np
import rtjlib as R
from pkg import P
dev=P.MiRtj()
pk=[]
for (w,h,Q,amp) in [(1920,1088,255,8),(1920,1088,255,8),(320,240,255,8),(640,368,128,8),(320,240,255,40),(1920,1088,200,20),(64,48,255,8)]:
    enc=R.OracleEncoder(w,h,Q); pk.append(enc.encode(R.synth_frame(w,h,len(pk),amp=amp)))
d_stream, po, pl, hdrs = dev.upload_packets(pk, align=1)
sizes=[(int(p[6])|int(p[7])<<8)*(int(p[8])|int(p[9])<<8)*3//2 for p in pk]
oo=np.concatenate([[0],np.cumsum([(s+255)//256*256 for s in sizes])]).astype(np.uint64)
d_out=dev.alloc(int(oo[-1])); dev.memset(d_out,0,int(oo[-1]))
plan=dev.plan(hdrs,po,pl,oo[:-1].copy())
plan.profile(True)
plan.decode(d_stream,d_out); dev.sync()
print(plan.times())
idx=plan.read_index(); k=0; dec=R.OracleDecoder()
for i,p in enumerate(pk):
    want=dec.block_offsets(p).astype(np.int64)-12
    got=idx[k:k+want.size].astype(np.int64); k+=want.size
    bad=np.nonzero(got!=want)[0]
    print(i, 'index ok' if bad.size==0 else ('BAD',bad[:5],got[bad[:5]],want[bad[:5]]))
    w_=np.zeros(sizes[i],np.uint8); R.OracleDecoder().decode(p,w_)
    g=dev.d2h(d_out,sizes[i],offset=int(oo[i])); print('   planes', np.array_equal(g,w_))
