#!/bin/bash
# build the library of another commit (default HEAD) as opticalflowdiffusion_amd/lib/libofd_hip_<tag>.so for same-box A/Bs (OFD_LIB=...)
set -e
REV=${1:-HEAD}; TAG=${2:-base}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/ofd_wt_$TAG && git -C "$ROOT" worktree prune && git -C "$ROOT" worktree add -f /tmp/ofd_wt_$TAG $REV > /dev/null 2>&1
(cd /tmp/ofd_wt_$TAG && python -m opticalflowdiffusion_amd.build > /dev/null)
cp /tmp/ofd_wt_$TAG/opticalflowdiffusion_amd/lib/libofd_hip.so "$ROOT/opticalflowdiffusion_amd/lib/libofd_hip_$TAG.so"
git -C "$ROOT" worktree remove --force /tmp/ofd_wt_$TAG
echo "built opticalflowdiffusion_amd/lib/libofd_hip_$TAG.so from $REV"
