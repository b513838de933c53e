for rep in 1 2; do for f in 0 1; do echo "## MMLF_ESE_FUSED=$f"; MMLF_ESE_FUSED=$f python tools/ese_bench.py 512 2>/dev/null | grep ESE; done; done
