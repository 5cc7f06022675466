#!/bin/bash
# The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (gcc; CPU only -- GPU sanitizers are not available on this pool).
# The oracle is what every parity claim is checked against: an out-of-bounds read or an uninitialised-looking value there would make a
# "match" meaningless.  Runs the CPU tests that drive the oracle on the sanitized build.   usage: tools/sanitize_oracle.sh [pytest args]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -s -C $R/oracle sanitize
export MOCCA_ORACLE_SANITIZED=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
cd $R
python -m pytest tests -q -m "not gpu" -k "oracle or golden or pybullet or host" -p no:cacheprovider "$@"
