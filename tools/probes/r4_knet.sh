cd /root/repo
timeout 600 python -m pytest tests/test_knet_body_gpu.py tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -15
for p in 1 0; do echo "== VIRNET_KNET_PERSISTENT=$p"; VIRNET_KNET_PERSISTENT=$p timeout 300 python tools/probes/sisr_latency.py 1 16 2>&1 | grep -v amdgpu; done
