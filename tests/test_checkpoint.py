"""npz weight interchange: names and layouts of the reference's TF variables."""
import types

import numpy as np
import pytest

from cutie_amd import checkpoint
from cutie_amd.model_cutie_aspp import CUTIERes


def _net():
    p = types.SimpleNamespace(embedding_size=128, weight_decay=5e-4, hard_negative_ratio=3, batch_size=1)
    return CUTIERes(500, 18, p)


def test_expected_variables_follow_the_reference_names():
    want = checkpoint.expected_variables(_net())
    names = list(want)
    assert names[0] == "embedding/weights" and want["embedding/weights"] == (500, 128)
    assert want["encoder1_1/weights"] == (3, 5, 128, 256) and want["encoder1_1/biases"] == (256,)
    assert want["aspp_1x1/weights"] == (1, 1, 1024, 256) and want["decoder1_1/weights"] == (3, 5, 512, 64)
    assert want["cls_logits/weights"] == (1, 1, 64, 18)
    assert len(names) == 1 + 2 * 14
    assert sum(int(np.prod(s)) for s in want.values()) == 13639890 - (20000 - 500) * 128      # SURVEY: 13,639,890 at V=20000


def test_round_trip_and_validation(tmp_path):
    net = _net()
    rs = np.random.RandomState(0)
    named = {k: rs.randn(*s).astype(np.float32) for k, s in checkpoint.expected_variables(net).items()}
    path = str(tmp_path / "w.npz")
    checkpoint.save_npz(path, named)
    back = checkpoint.load_npz(path, net)
    assert list(back) == list(checkpoint.expected_variables(net))
    assert all(np.array_equal(back[k], named[k]) for k in named)

    extra = dict(named)
    extra["encoder1_1/weights/Adam"] = np.zeros((3, 5, 128, 256), np.float32)
    checkpoint.save_npz(path, extra)
    with pytest.raises(KeyError):
        checkpoint.load_npz(path, net)
    assert "encoder1_1/weights/Adam" not in checkpoint.load_npz(path, net, strict=False)

    short = {k: v for k, v in named.items() if k != "aspp_2/biases"}
    checkpoint.save_npz(path, short)
    with pytest.raises(KeyError, match="aspp_2/biases"):
        checkpoint.load_npz(path, net)

    bad = dict(named)
    bad["decoder1_1/weights"] = np.zeros((5, 3, 512, 64), np.float32)      # OIHW / transposed kernels are refused
    checkpoint.save_npz(path, bad)
    with pytest.raises(ValueError, match="decoder1_1/weights"):
        checkpoint.load_npz(path, net)


@pytest.mark.gpu
def test_engine_loads_an_npz(tmp_path):
    import torch
    from cutie_amd.runtime import Engine
    net = _net()
    eng = Engine(net, "cuda:0", dtype="f32", seed=3)
    path = str(tmp_path / "w.npz")
    checkpoint.save_npz(path, eng.state_dict())
    grid = torch.randint(0, 500, (1, 16, 32), dtype=torch.int32)
    want = eng.get(eng.forward(grid), "softmax").cpu().clone()
    other = Engine(net, "cuda:0", dtype="f32", seed=99)
    assert not torch.equal(other.get(other.forward(grid), "softmax").cpu(), want)
    other.load_params(checkpoint.load_npz(path, net))
    assert torch.allclose(other.get(other.forward(grid), "softmax").cpu(), want, atol=2e-4)      # run-to-run noise of the atomically summed norm moments
