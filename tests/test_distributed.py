"""CPU, world_size 2 over gloo: the data-parallel path (utterance shards, sync-BN statistic hand-off, one
all-reduce of the flat gradient arena) reproduces the single-process global-batch step.  On the GPU box the same
code runs over RCCL (backend "nccl"); kernels here come from the emulator build (test infrastructure)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import numpy_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "_build", "libtcr_emu.so")


def _setup(lib):
    import tcresnet_amd as T
    arch = R.make_tcresnet("TCResNet8", 1.0)
    p, s = R.init_params(arch, 0)
    R.randomize_bn(arch, p, s)
    fe = T.Frontend(window_size_samples=640, window_stride_samples=320, lib=lib)
    net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, lib=lib)
    sd = dict(p); sd.update(s)
    net.load_state_dict(sd)
    return fe, net


def _worker(rank, world, port, sync_bn, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    import tcresnet_amd as T
    from tcresnet_amd.parallel import DataParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = T._lib.load_from(EMU)
    fe, net = _setup(lib)
    b = 3
    wav = torch.from_numpy(R.synth_waveforms(b, seed=77, start=rank * b))
    lab = torch.from_numpy(R.synth_labels(b, start=rank * b))
    dp = DataParallel(net, sync_bn=sync_bn)
    assert dp.enabled and dp.world == world and dp.rank == rank
    logits, probs, loss_sum = dp.forward_train(fe(wav), lab, keep_prob=0.5, seed=3)
    g = dp.backward()
    mean_loss = dp.mean_loss(loss_sum, b)
    net.sgd_momentum_step(0.1, 0.9, 0.001)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), grads=g.numpy(), logits=logits.numpy(), loss=float(mean_loss),
             params=net.params.numpy(), stats=net.stats.numpy())
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("sync_bn", [True, False])
def test_two_replicas_match_global_batch(emu_lib, tmp_path, sync_bn):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), sync_bn, str(tmp_path)), nprocs=world, join=True)
    r = [dict(np.load(tmp_path / f"rank{i}.npz")) for i in range(world)]
    # replicas hold identical gradients / parameters / moving statistics after the all-reduce
    assert np.array_equal(r[0]["grads"], r[1]["grads"]) and np.array_equal(r[0]["params"], r[1]["params"])
    assert abs(float(r[0]["loss"]) - float(r[1]["loss"])) == 0.0
    if sync_bn:
        assert np.array_equal(r[0]["stats"], r[1]["stats"])
    # single process, global batch of 6, same dropout stream (masks are indexed by global sample id)
    fe, net = _setup(emu_lib)
    wav = torch.from_numpy(R.synth_waveforms(6, seed=77))
    lab = torch.from_numpy(R.synth_labels(6))
    logits, probs, loss_sum = net.forward_train(fe(wav), lab, keep_prob=0.5, seed=3)
    g = net.backward().numpy().copy()
    if sync_bn:
        # cross-replica statistics == the reference's single-device global-batch BN: everything matches
        assert np.abs(np.concatenate([r[0]["logits"], r[1]["logits"]]) - logits.numpy()).max() < 2e-5
        assert abs(float(r[0]["loss"]) - float(loss_sum) / 6) < 1e-5
        assert np.abs(r[0]["grads"] - g).max() < 2e-5 * max(1.0, np.abs(g).max())
        net.sgd_momentum_step(0.1, 0.9, 0.001)
        assert np.abs(r[0]["params"] - net.params.numpy()).max() < 1e-5
        assert np.abs(r[0]["stats"] - net.stats.numpy()).max() < 1e-5
    else:
        # per-replica BN is the documented deviation: finite, but not the global-batch statistics
        assert np.isfinite(r[0]["grads"]).all() and np.abs(r[0]["grads"] - g).max() > 1e-4
