set -e
for L in ${LIBS:-B}; do
echo "== lib $L wide table, bitmap step"
NIMBLE_AMD_LIB=tools/bin/libnimble_amd_phase_$L.so PHASE_LAUNCH=${PHASE_LAUNCH:-1} timeout -k 10 170 python tools/phase_timing.py wide 0 5000
done
