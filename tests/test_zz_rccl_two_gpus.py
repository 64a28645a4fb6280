"""The RCCL (backend "nccl") forms of the two multi-process tests on the real kernels: one GPU per rank, device-to-device
point-to-point transfers over xGMI.  They need two GPUs, so they SKIP on the one-GPU test box - this path has not run on
hardware yet.  Kept in the last test file on purpose: on a multi-GPU box their first contact with RCCL
happens after every other test has reported."""
import pytest
import torch

pytestmark = pytest.mark.gpu

needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL path needs two GPUs")


@needs_two
def test_encode_files_two_ranks_over_rccl(tmp_path):
    from test_gpu_round2 import test_encode_files_two_ranks_real_kernels as run
    run.__wrapped__(tmp_path, "nccl", False) if hasattr(run, "__wrapped__") else run(tmp_path, "nccl", False)


@needs_two
def test_one_clip_split_over_two_ranks_over_rccl(tmp_path):
    from test_gpu_round3 import test_one_clip_split_over_ranks_real_kernels as run
    run.__wrapped__(tmp_path, "nccl", 2) if hasattr(run, "__wrapped__") else run(tmp_path, "nccl", 2)
