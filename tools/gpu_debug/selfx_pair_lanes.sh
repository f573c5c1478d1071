for r in 1 2; do
for v in 1 0; do echo "LOOP=0 MT_LANES=$v 512x64"; ROMS_HIP_LOOP=0 ROMS_HIP_MT_LANES=$v timeout 100 python tools/gpu_debug/gpu_selfx_prof.py benchmark1 40 peer 2>&1 | tail -1; done
done
