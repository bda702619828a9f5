set -u
mkdir -p gpurun_out/r02g
for w in 1 2 4 8; do tools/ubench/valu_snop $w > gpurun_out/r02g/valu_snop_${w}w.txt; done
paste -d'|' gpurun_out/r02g/valu_snop_1w.txt gpurun_out/r02g/valu_snop_2w.txt gpurun_out/r02g/valu_snop_4w.txt gpurun_out/r02g/valu_snop_8w.txt | awk -F'|' '{split($1,a," ns"); split($2,b," ns"); split($3,c," ns"); split($4,e," ns"); n=split(a[1],x," "); m=split(b[1],y," "); o=split(c[1],z," "); q=split(e[1],u," "); printf "%-38s 1w %s 2w %s 4w %s 8w %s\n", substr($1,1,38), x[n], y[m], z[o], u[q]}'
