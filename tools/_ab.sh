python -m pytest tests/test_gpu_block_fused.py tests/test_gpu_bf16.py tests/test_gpu_fp8.py -q -x 2>&1 | tail -3
for rep in 1 2; do for v in 0 1; do echo "== MICF_ATTN_BWD_VALU=$v"; MICF_ATTN_BWD_VALU=$v python tools/bench_block.py --dtype bf16 --fused-only 2>/dev/null | cut -c1-170; MICF_ATTN_BWD_VALU=$v python tools/bench_block.py --dtype bf16 --fused-only --cross 2>/dev/null | cut -c1-170; done; done
bash tools/ab_envs.sh abs4 - "MICF_ATTN_BWD_VALU=1"
