# ablations of the one-wave inverse (ubench only): 1 no off-chain MFMAs, 2 no LDS broadcast, 3 no 4x4 factorisation
O=gpurun_out/r4d; mkdir -p $O; : > $O/abl.txt
for a in 0 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCT_IW_ABL=$a scripts/ubench/inv_wave.hip -o /tmp/inv_wave_$a 2>> $O/build.err
  echo "CT_IW_ABL=$a" >> $O/abl.txt; timeout 120 /tmp/inv_wave_$a 2>&1 | grep "ticks per inverse" >> $O/abl.txt
done
