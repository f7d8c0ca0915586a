#!/bin/bash
# A/B of two builds on one box: libhp_old.so (a build of another revision, copied next to the library) vs the current one
P=3d-point-clouds-autocomplete_amd/hyperpocket_amd
cp $P/libhyperpocket_hip.so /tmp/hp_new.so
for i in 1 2; do
  cp $P/libhp_old.so $P/libhyperpocket_hip.so; echo OLD; "$@"
  cp /tmp/hp_new.so $P/libhyperpocket_hip.so; echo NEW; "$@"
done
