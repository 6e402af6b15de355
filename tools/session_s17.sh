set -o pipefail
NCCL_DEBUG=INFO tools/step.sh s17_rccl --timeout 300 -- python -m pytest tests/test_gpu_shard.py::test_engine_rccl_reduction_world_size_one -x -q -s
