#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root/tools
for b in wgx_*; do echo "== $b"; timeout -k 10 60 ./$b | sed 's/(kernel.*split;//' ; done
