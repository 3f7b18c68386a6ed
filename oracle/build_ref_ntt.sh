#!/bin/bash
# ORACLE build (test infrastructure): the reference's own NTT kernel, compiled from the sources where they lie under
# /root/reference — dot_ring/ring_proof/polynomial/ntt.pyx and dot_ring/curve/native_field/scalar.pyx (Cython) over
# bls12_381_scalar.c — with every output under oracle/_ref/ (git-ignored).  Nothing is copied, no stand-in is written:
# cython is told where to put the generated C (-o), gcc reads the reference's .c / .h in place, and the two extension modules
# land in oracle/_ref/pyx/dot_ring/... as an implicit namespace package (no __init__.py), importable with oracle/_ref/pyx on
# sys.path.  Used only by oracle/gen_ntt_fixtures.py (which writes tests/golden/ntt/) and tests/test_oracle_kats.py.
set -euo pipefail
here=$(cd "$(dirname "$0")" && pwd)
ref=${REFERENCE_ROOT:-/root/reference}
if [ ! -f "$ref/dot_ring/ring_proof/polynomial/ntt.pyx" ]; then
  echo "reference sources absent: keeping prebuilt oracle/_ref/pyx (if any)"
  exit 0
fi
out=$here/_ref/pyx
gen=$here/_ref/build
nf=$ref/dot_ring/curve/native_field
mkdir -p "$gen" "$out/dot_ring/curve/native_field" "$out/dot_ring/ring_proof/polynomial"
suffix=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
inc=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
python3 -m cython -3 -I "$ref" -o "$gen/scalar.c" "$nf/scalar.pyx"
python3 -m cython -3 -I "$ref" -o "$gen/ntt.c" "$ref/dot_ring/ring_proof/polynomial/ntt.pyx"
gcc -O2 -fPIC -shared -I "$inc" -I "$nf" "$gen/scalar.c" "$nf/bls12_381_scalar.c" -o "$out/dot_ring/curve/native_field/scalar$suffix"
gcc -O2 -fPIC -shared -I "$inc" -I "$nf" "$gen/ntt.c" "$nf/bls12_381_scalar.c" -o "$out/dot_ring/ring_proof/polynomial/ntt$suffix"
echo "built oracle/_ref/pyx (reference ntt.pyx + scalar.pyx)"
