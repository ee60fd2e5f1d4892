#!/bin/bash
# Micro-benchmark evidence cited by profiles/README.md: run through gpurun, then copy gpurun_out/r03_*.txt to profiles/.
set -e
mkdir -p gpurun_out
timeout -k 10 400 python tools/conv3x3_bench.py all 3 > gpurun_out/r03_conv3x3_bench.txt 2>&1
echo "conv3x3 bench done"
{ for lt in layer1:1 layer3:0 fusion1_c3:0; do echo "== ${lt%:*} (tile ${lt#*:})"; timeout -k 10 120 python tools/conv3x3_stamps.py ${lt%:*} ${lt#*:}; done; } > gpurun_out/r03_conv3x3_stamps.txt 2>&1
echo "stamps done"
{ timeout -k 10 200 python tools/frontend_bench.py 3
  # the A/B library = this tree's objects + the round-2 voxeliser (build/old/voxelize_r2.o: hipcc -c of `git show f202bdd:.../csrc/voxelize.hip`);
  # relinked here so that it always exports the current symbol set
  if [ -f build/old/voxelize_r2.o ]; then
    hipcc --offload-arch=gfx950 -shared -fPIC -o build/old/libbevf_oldvox.so \
      $(ls bevfusion_multimodal_3d_object_detection_amd/csrc/*.o | grep -v "/voxelize.o") build/old/voxelize_r2.o
  fi
  if [ -f build/old/libbevf_oldvox.so ]; then
    echo "== round-2 voxeliser (BEVF_AB_LIB=build/old/libbevf_oldvox.so)"
    BEVF_AB_LIB=build/old/libbevf_oldvox.so timeout -k 10 200 python tools/frontend_bench.py 3
  fi; } > gpurun_out/r03_frontend_bench.txt 2>&1
echo "frontend done"
{ timeout -k 10 200 python tools/wino_bench.py
  for l in layer1 layer1r layer2 layer3 fusion1; do timeout -k 10 100 python tools/wino_stamps.py $l 0; done; } > gpurun_out/r03_wino_bench.txt 2>&1
echo "wino done"
timeout -k 10 200 python tools/stem_bench.py 48 > gpurun_out/r03_stem_bench.txt 2>&1
echo "stem done"
