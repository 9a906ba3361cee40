#!/bin/bash
# The CPU oracle under AddressSanitizer + UBSan (the GPU pool refuses sanitizer runs: this is the sanitizer coverage of the C side that
# shares the model tables, the state layout and the config struct with the kernels).  Runs the oracle's own CPU tests on the
# instrumented build:  tools/oracle_sanitize.sh [pytest args]      -> profiles/<tag>_oracle_sanitize.txt by the caller
cd "$(dirname "$0")/.."
make -s -C oracle sanitize || exit 1
export REX_ORACLE_SANITIZE=1 ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
exec python -m pytest tests/test_oracle_controller.py tests/test_oracle_env_commands.py tests/test_oracle_physics.py tests/test_oracle_rollouts.py tests/test_sharding_gloo.py -q -m "not gpu" -p no:cacheprovider "$@"
