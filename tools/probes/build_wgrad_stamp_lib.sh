# Instrumented build of the library for tools/probes/wgrad_stamps.py: csrc/libW.so (never shipped, never loaded by default)
set -e
cd "$(dirname "$0")/../../openseg3d_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -DSEG3D_WGRAD_STAMP -c wgrad_split.hip -o /tmp/wgrad_split_stamp.o
objs=$(ls *.o | grep -v '^wgrad_split.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libW.so $objs /tmp/wgrad_split_stamp.o
ls -la libW.so
