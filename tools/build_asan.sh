#!/bin/bash
# Host-side AddressSanitizer + UBSan variant of libffk.so: host code instrumented,
# device code untouched (-fno-gpu-sanitize; GPU ASan is not available on this pool), allocations of
# the arena / block pools stubbed to the C heap (-DFFK_HOST_SANITIZE) so that their logic runs
# without a GPU.  Output: build/libffk_asan.so (never shipped).  Run the checks with
#   python -m pytest tests/test_sanitizer_build.py -q        (or: tools/run_asan_checks.sh)
set -e
cd "$(dirname "$0")/.."
make -C filter_functions_amd/csrc -j"${JOBS:-6}" VARIANT=asan \
  VFLAGS="-fsanitize=address,undefined -fno-gpu-sanitize -shared-libasan -fno-omit-frame-pointer -g -O1 -DFFK_HOST_SANITIZE" \
  HIPCC_LINK_FLAGS="-fsanitize=address,undefined -shared-libasan"
