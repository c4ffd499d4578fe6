#!/bin/bash
# On the GPU box: the phase profile of the paired grouped stream with the -DPDS_PROFILE_MID variant, then the default library back.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; LIB=$ROOT/polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
cp $LIB /tmp/default.so; trap 'cp /tmp/default.so $LIB' EXIT
cp $ROOT/tools/variants/prof.bin $LIB
python $ROOT/tools/grouped_mid_profile.py
