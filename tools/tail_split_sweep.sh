#!/bin/bash
# where the second workgroup of a channel starts (percent of the push; DH_TAIL_SPLIT overrides the engine's default): the bench workload per setting
for v in ${*:-0 70 75 80 85 90 "75,93"}; do echo "== DH_TAIL_SPLIT=$v"; DH_TAIL_SPLIT=$v python tools/lib_ab.py ${PROTO:-dmr} digiham_amd/libdigiham_amd.so 2>&1 | tail -2 | cut -c1-110; done
