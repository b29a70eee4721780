# A/B on one box: conv fabric traffic with round 4's spconv_tile.hip linked into today's library vs the shipped library
C=openseg3d_amd/csrc
cp $C/libseg3d_hip.so $C/keep.so
cp $C/libR4tile.so $C/libseg3d_hip.so && bash tools/collect_traffic.sh r4tile; rc=$?
cp $C/keep.so $C/libseg3d_hip.so
[ $rc = 0 ] || exit $rc
bash tools/collect_traffic.sh r5tile
