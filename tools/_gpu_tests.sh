mkdir -p gpurun_out/r2f
rm -f gpurun_out/r2f/parity.txt
TSSPLAT_AMD_PARITY_REPORT=gpurun_out/r2f/parity.txt python -m pytest tests -m gpu -q 2>&1 | tail -25
