mkdir -p gpurun_out/r06c
for ct in 16 32 64; do
  MPNN_CONV_CT=$ct MPNN_CONV_DBG=0 timeout 200 python tools/ablate_saturated.py 1024 > gpurun_out/r06c/ct_$ct.txt 2>&1
done
paste gpurun_out/r06c/ct_16.txt gpurun_out/r06c/ct_32.txt gpurun_out/r06c/ct_64.txt | grep "^op" | awk -F'\t' '{printf "%-12s %-22s %8s %8s %8s\n",$2,$3,$4,$8,$12}'
