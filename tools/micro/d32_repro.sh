#!/bin/bash
# The round-3 d = 32 failure (rc2_reduce1<double, 32> with 444 B of scratch per lane: lanes 5, 6, 7 of a chain come out of
# F C F^T + Q with a zero row; profiles/r04_experiments.txt item 1), reproducible on demand -- e.g. against a new compiler drop:
#
#   tools/micro/d32_repro.sh build     where the repository's history is (no GPU needed): checks commit edfddd1 out into a
#                                      temporary worktree, builds its library with the hipcc on PATH and leaves it as
#                                      tools/micro/libpgps_edfddd1.so (git-ignored; it travels to the GPU box with gpurun)
#   tools/micro/d32_repro.sh run       on the GPU box: tools/d32_probe.py (closed-form chains) through that library at d = 32
#                                      -- a compiler that is fine prints "exact" everywhere; the failure prints the map of
#                                      wrong rows (5, 6, 7) -- and through the shipped library for comparison
# Not one kernel in 300 lines: the kernel is 8.6 k lines of generated code whose failure vanished with unrelated changes of
# its loads (profiles/r04_experiments.txt), so the reproducer is the commit itself.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
case "$1" in
build)
    W=$(mktemp -d /tmp/pgps_edfddd1.XXXXXX)
    git -C "$R" worktree add -f "$W" edfddd1 > /dev/null
    make -C "$W/parallel-gps_amd/csrc" -j"$(nproc)" > "$W/build.log" 2>&1 || { tail -20 "$W/build.log"; exit 1; }
    cp "$W/parallel-gps_amd/pssgp/libpgps.so" "$R/tools/micro/libpgps_edfddd1.so"
    git -C "$R" worktree remove --force "$W"
    sha256sum "$R/tools/micro/libpgps_edfddd1.so"
    ;;
run)
    cd "$R"
    echo "== library of commit edfddd1"
    PGPS_LIB=$R/tools/micro/libpgps_edfddd1.so python tools/d32_probe.py 32 96 16
    echo "== shipped library"
    python tools/d32_probe.py 32 96 16
    ;;
*)  echo "usage: $0 build | run"; exit 2;;
esac
