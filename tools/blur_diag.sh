# Development aid (runs on the GPU box, in the scratch copy of the repo): compile-time variants of the fused blur + response kernel.
cd $GRAFT_REPO_ROOT/mods-light-zmq_amd
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
for V in "-DBLUR_DIAG=0" "-DBLUR_DIAG=4" "-DBLUR_DIAG=8" "-DBLUR_DIAG=12" "-DBLUR_DIAG=15"; do
  /opt/rocm/bin/hipcc $FLAGS $V -c csrc/pyramid.hip -o csrc/pyramid.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmodsgpu.so csrc/*.o -L/opt/rocm/lib -lrccl
  echo "variant $V"; python ../tools/prof_detect.py 16 2>&1 | grep "^blur"
done
