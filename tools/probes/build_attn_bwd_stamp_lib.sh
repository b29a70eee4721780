# Instrumented build of the library for tools/probes/attn_bwd_stamps.py: csrc/libS.so (never shipped, never loaded by default)
set -e
cd "$(dirname "$0")/../../openseg3d_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -DSEG3D_ATTN_STAMP -c attention_fused_bwd.hip -o /tmp/attention_fused_bwd_stamp.o
objs=$(ls *.o | grep -v '^attention_fused_bwd.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libS.so $objs /tmp/attention_fused_bwd_stamp.o
ls -la libS.so
