"""Pins oracle/cyclepose_ref.py against outputs of the reference's own ConditionalGenerator
(MPEM/architecture_v3.py:195-226) and PoseOperator (UTILS/geometry_utils.py:230-265), produced by
oracle/make_golden.py."""
import os

import numpy as np
import pytest
import torch

from oracle import cyclepose_ref as CP


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "cyclepose_pose.npz"))


def _input(seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.uniform(-1, 1, size=(3, 6, 128, 128)).astype(np.float32))


def test_pose_matches_reference(gold):
    w = CP.synth_weights(int(gold["weight_seed"]))
    x = _input(int(gold["input_seed"]))
    taps = {}
    with torch.no_grad():
        T = CP.pose_matrix(CP.pose7(w, x, taps))
    assert T.shape == (3, 4, 4) and T.dtype == torch.float32
    assert np.allclose(T.numpy(), gold["T"], rtol=0, atol=2e-6)
    assert np.allclose(taps["c0"].mean(dim=(2, 3)).numpy(), gold["c0_mean"], atol=1e-6)
    assert np.allclose(taps["c0"][:, :, ::16, ::16].numpy(), gold["c0_sample"], atol=2e-5)
    assert np.allclose(taps["c2"][:, ::8, ::4, ::4].numpy(), gold["c2_sample"], atol=2e-5)
    assert np.allclose(taps["pooled"].numpy(), gold["pooled"], atol=2e-5)


def test_pose_is_se3(gold):
    T = gold["T"]
    for m in T:
        assert np.allclose(m[:3, :3] @ m[:3, :3].T, np.eye(3), atol=1e-5)
        assert np.array_equal(m[3], np.array([0, 0, 0, 1], np.float32))


def test_quaternion_ops(gold):
    q = torch.from_numpy(gold["quat_in"])
    qn = q / torch.norm(q, p=2, dim=-1, keepdim=True)
    assert np.array_equal(qn.numpy(), gold["quat_norm"])
    assert np.allclose(CP.quaternion_to_matrix(qn).numpy(), gold["quat_R"], atol=1e-7)


def test_center_crop_offsets():
    # torchvision CenterCrop(128) on 640x480 -> top 176, left 256 (mpem_interface.py:40-44)
    f = torch.zeros(2, 480, 640, 3, dtype=torch.uint8)
    f[0, 176, 256] = 255
    f[1, 176 + 127, 256 + 127] = 255
    x = CP.center_crop_pair(f, torch.tensor([[0, 1]]))
    assert x.shape == (1, 6, 128, 128)
    assert x[0, 0, 0, 0] == 1.0 and x[0, 3, 127, 127] == 1.0
    assert x[0, 0, 1, 1] == -1.0
