# Instrumented build of the library for tools/probes/conv_stamps.py: csrc/libS.so (never shipped, never loaded by default)
set -e
cd "$(dirname "$0")/../../openseg3d_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -DSEG3D_CONV_STAMP -c spconv_split.hip -o /tmp/spconv_split_stamp.o
objs=$(ls *.o | grep -v '^spconv_split.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libS.so $objs /tmp/spconv_split_stamp.o
ls -la libS.so
