#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== sim-world 8"; tools/variant_ab.sh --sim-world 8 --steps 3 --warmup 1 --other-configs off -- default drop2 drop8 drop24
echo "== full"; tools/variant_ab.sh --steps 2 --warmup 1 --other-configs off -- default drop2 drop8 drop24
echo "== C4 spp 32"; tools/variant_ab.sh --config C4 --spp 32 --steps 1 --warmup 1 --other-configs off -- default drop8
echo "== C3 spp 128"; tools/variant_ab.sh --config C3 --spp 128 --steps 1 --warmup 1 --other-configs off -- default drop8
