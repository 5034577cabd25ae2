"""world_size 2: the data-parallel path (utterance shards, sync-BN statistic hand-off, one all-reduce of the flat gradient
arena) reproduces the single-process global-batch step.

  * CPU (`-m "not gpu"`): gloo, kernels from the emulator build (test infrastructure);
  * GPU (`-m gpu`): the SAME code with the gfx950 library -- two processes time-sharing the one GPU of the test box over gloo
    (the box has a single device, so RCCL itself cannot be exercised there; on the 8-GPU node the identical calls go through
    backend "nccl"), TCResNet14-1.5 = BASELINE.json configs[3]'s model."""
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


def _lib_of(kind):
    import tcresnet_amd as T
    return T._lib.load_from(EMU) if kind == "emu" else T._lib.get()


def _setup(lib, name="TCResNet8", width=1.0):
    import tcresnet_amd as T
    arch = R.make_tcresnet(name, width)
    p, s = R.init_params(arch, 0)
    R.randomize_bn(arch, p, s)
    dev = "cuda" if lib.kind == "hip" else None
    fe = T.Frontend(window_size_samples=640, window_stride_samples=320, lib=lib, device=dev)
    net = T.TCResNet(name, R.tcresnet_channels(name, width), 40, fe.n_frames, 12, lib=lib, device=dev)
    sd = dict(p); sd.update(s)
    net.load_state_dict(sd)
    return fe, net


def _setup_dscnn(lib, size="S"):
    import tcresnet_amd as T
    from oracle import dscnn_ref as D
    p, s = D.init_params(D.net_def(size), seed=2)
    dev = "cuda" if lib.kind == "hip" else None
    fe = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, lib=lib, device=dev)
    net = T.DSCNN(size, fe.n_frames, 10, 12, lib=lib, device=dev)
    sd = dict(p); sd.update(s)
    net.load_state_dict(sd)
    return fe, net


def _setup_graph(lib, variant):
    """A graph-engine family (Res15Narrow: 19 channels, dilated 3 x 3 convs, BN without scale) on a small 16 x 10 feature plane."""
    import tcresnet_amd as T
    from tcresnet_amd import runtime
    from tcresnet_amd.audio_nets import res, tc_resnet
    dev = "cuda" if lib.kind == "hip" else "cpu"
    runtime.set_default(lib, dev)
    tc_resnet.reset_engines()
    fe = T.Frontend(window_size_samples=960, window_stride_samples=960, num_mfccs=10, lib=lib, device=None if dev == "cpu" else dev)
    net = res.get_engine(variant, fe.n_frames, 10, 12)
    net.init_variables(3)
    rng = np.random.RandomState(4)
    sd = {k: v for k, v in net.state_dict().items() if k in net.tensors}
    for k, v in sd.items():
        if net.tensors[k].kind in (1, 4):
            sd[k] = rng.uniform(0.5, 1.5, v.shape).astype(np.float32)
        elif net.tensors[k].kind in (2, 3):
            sd[k] = rng.uniform(-0.5, 0.5, v.shape).astype(np.float32)
    net.load_state_dict(sd)
    return fe, net


def _make(lib, name, width):
    if name == "DSCNN":
        return _setup_dscnn(lib, width)
    if name.startswith("Res"):
        return _setup_graph(lib, name)
    return _setup(lib, name, width)


def _worker(rank, world, port, sync_bn, out_dir, kind, name, width, b):
    import sys
    sys.path.insert(0, ROOT)
    from tcresnet_amd.parallel import DataParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _lib_of(kind)
    fe, net = _make(lib, name, width)
    wav = torch.from_numpy(R.synth_waveforms(b, seed=77, start=rank * b)).to(fe.device)
    lab = torch.from_numpy(R.synth_labels(b, start=rank * b)).to(fe.device)
    dp = DataParallel(net, sync_bn=sync_bn)
    assert dp.enabled and dp.world == world and dp.rank == rank
    logits, probs, loss_sum = dp.forward_train(fe(wav), lab, keep_prob=0.5, seed=3)
    g = dp.backward()
    mean_loss = dp.mean_loss(loss_sum, b)
    net.sgd_momentum_step(0.1, 0.9, 0.001)
    torch.cuda.synchronize() if fe.device.type == "cuda" else None
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), grads=g.cpu().numpy(), logits=logits.cpu().numpy(), loss=float(mean_loss),
             params=net.params.cpu().numpy(), stats=net.stats.cpu().numpy(), collectives=dp.collectives,
             handoffs=(2 * lib.tcr_dscnn_num_units(net._h) if name == "DSCNN" else 2 * (lib.tcr_g2d_num_stages(net._h) - 1) if name.startswith("Res")
                       else lib.tcr_net_num_levels(net._h, 0) + lib.tcr_net_num_levels(net._h, 1) - 2))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _two_replicas(lib, kind, tmp_path, sync_bn, name, width, b, grad_tol=2e-5, world=2):
    mp.spawn(_worker, args=(world, _free_port(), sync_bn, str(tmp_path), kind, name, width, b), nprocs=world, join=True)
    r = [dict(np.load(tmp_path / f"rank{i}.npz")) for i in range(world)]
    # replicas hold identical gradients / parameters / moving statistics after the all-reduce
    for ri in r[1:]:
        assert np.array_equal(r[0]["grads"], ri["grads"]) and np.array_equal(r[0]["params"], ri["params"])
        assert abs(float(r[0]["loss"]) - float(ri["loss"])) == 0.0
    # collectives per step: ONE for the gradient arena (the loss sum rides in its tail), plus -- cross-replica BN only -- one per
    # dependency level in the forward and one in the backward: TC-ResNet 2 x (1 + 2 x blocks) (TCResNet8: 14, TCResNet14: 26; a block's
    # shortcut unit rides with the conv of its level), DS-CNN one per BN unit (2 x 11)
    for ri in r:
        assert int(ri["collectives"]) == 1 + (int(ri["handoffs"]) if sync_bn else 0), (int(ri["collectives"]), int(ri["handoffs"]))
        if name == "TCResNet8":
            assert int(ri["handoffs"]) == 14
        elif name == "TCResNet14":
            assert int(ri["handoffs"]) == 26
    if sync_bn:
        for ri in r[1:]:
            assert np.array_equal(r[0]["stats"], ri["stats"])
    # single process, global batch of world x b, same dropout stream (masks are indexed by global sample id)
    fe, net = _make(lib, name, width)
    wav = torch.from_numpy(R.synth_waveforms(world * b, seed=77)).to(fe.device)
    lab = torch.from_numpy(R.synth_labels(world * b)).to(fe.device)
    logits, probs, loss_sum = net.forward_train(fe(wav), lab, keep_prob=0.5, seed=3)
    g = net.backward().cpu().numpy().copy()
    if sync_bn:
        # cross-replica statistics == the reference's single-device global-batch BN: everything matches
        assert np.abs(np.concatenate([ri["logits"] for ri in r]) - logits.cpu().numpy()).max() < 2e-5
        assert abs(float(r[0]["loss"]) - float(loss_sum) / (world * b)) < 1e-5
        assert np.abs(r[0]["grads"] - g).max() < grad_tol * max(1.0, np.abs(g).max())
        net.sgd_momentum_step(0.1, 0.9, 0.001)
        assert np.abs(r[0]["params"] - net.params.cpu().numpy()).max() < max(1e-5, 0.1 * grad_tol * max(1.0, np.abs(g).max()))
        assert np.abs(r[0]["stats"] - net.stats.cpu().numpy()).max() < 1e-5
    else:
        # per-replica BN is the documented deviation: finite, but not the global-batch statistics
        assert np.isfinite(r[0]["grads"]).all() and np.abs(r[0]["grads"] - g).max() > 1e-4


@pytest.mark.parametrize("sync_bn", [True, False])
def test_two_replicas_match_global_batch(emu_lib, tmp_path, sync_bn):
    _two_replicas(emu_lib, "emu", tmp_path, sync_bn, "TCResNet8", 1.0, 3)


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("sync_bn", [True, False])
def test_many_replicas_match_global_batch(emu_lib, tmp_path, sync_bn, world):
    """BASELINE configs[3]'s layout is EIGHT ranks: sample offsets (rank x b), dropout indices by global sample, the 1 / global_batch
    scale, the level-by-level statistic hand-off and the moving-statistics update at world_size 4 and 8 (emulator, gloo) against the
    single-process global batch."""
    _two_replicas(emu_lib, "emu", tmp_path, sync_bn, "TCResNet8", 1.0, 2, world=world)


def test_four_replicas_match_global_batch_tcresnet14(emu_lib, tmp_path):
    """configs[3]'s model (TCResNet14-1.5: identity and shortcut blocks, 26 hand-offs) at four ranks with cross-replica statistics
    (eight ranks: TCResNet8 above and bench.py --gpus 8 below -- this net at eight emulator processes alone takes ~1.5 min of the suite)."""
    _two_replicas(emu_lib, "emu", tmp_path, True, "TCResNet14", 1.5, 1, world=4)


def test_two_replicas_match_global_batch_graph_engine(emu_lib, tmp_path):
    """A graph-engine family with cross-replica BN statistics (tcr_g2d_*_stage): two replicas == the single-device global batch."""
    from tcresnet_amd import runtime
    from tcresnet_amd.audio_nets import tc_resnet
    try:
        _two_replicas(emu_lib, "emu", tmp_path, True, "Res15Narrow", None, 2, grad_tol=5e-5)
    finally:
        runtime.set_default(None, None)
        tc_resnet.reset_engines()


def test_two_replicas_match_global_batch_dscnn(emu_lib, tmp_path):
    """DS-CNN with cross-replica BN statistics (tcr_dscnn_*_stage): two replicas == the single-device global batch."""
    _two_replicas(emu_lib, "emu", tmp_path, True, "DSCNN", "S", 2)


@pytest.mark.gpu
@pytest.mark.parametrize("size", ["S", "L"])
def test_two_replicas_match_global_batch_dscnn_hip(hip_lib, tmp_path, size):
    # The replicas add up the same partial rows in a different grouping than the single process, so scale / shift differ in the last
    # bit and a ReLU input within ~1e-7 of zero can land on the other side: DS-CNN-L has 6 M ReLU inputs at this batch and statistics
    # over 2 000 positions, so ONE such flip moves gradient entries by ~5e-4 of the largest (seen: 3.5e-4 .. 3.6e-3 with different
    # summation orders of the statistics).  S (0.6 M inputs) is held to the tight bound; L to 8e-3 with tight logits / loss.  (Staged == unstaged at one replica is bitwise for both: test_gpu_parity.)
    _two_replicas(hip_lib, "hip", tmp_path, True, "DSCNN", size, 8, grad_tol=2e-5 if size == "S" else 8e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("sync_bn", [True, False])
def test_two_replicas_match_global_batch_hip(hip_lib, tmp_path, sync_bn):
    """TCResNet14-1.5 (configs[3]) on the gfx950 library: staged sync-BN + arena all-reduce across two processes."""
    _two_replicas(hip_lib, "hip", tmp_path, sync_bn, "TCResNet14", 1.5, 16)


def _cli_worker(rank, world, port, kind, train_dir, sync_bn):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LOCAL_WORLD_SIZE=str(world), TCR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from tcresnet_amd import runtime, train_audio
    from tests.test_boundary import REF_TRAIN_CMD
    runtime.set_default(_lib_of(kind), "cpu" if kind == "emu" else "cuda")
    cmd = REF_TRAIN_CMD.replace("--batch_size 6", "--batch_size 3").replace("--max_step_from_restore 3", "--max_step_from_restore 2")
    if sync_bn:
        cmd = cmd.replace("--optimizer mom", "--sync_bn --optimizer mom")
    tr = train_audio.train(train_audio.parse_arguments(cmd.format(d=train_dir).split()))
    np.savez(os.path.join(train_dir, f"final{rank}.npz"), params=tr.model.engine.params.cpu().numpy(), stats=tr.model.engine.stats.cpu().numpy(),
             step=tr.global_step)
    dist.destroy_process_group()


def _cli_two_ranks(kind, tmp_path, sync_bn):
    from tcresnet_amd.common import tf_bundle
    mp.spawn(_cli_worker, args=(2, _free_port(), kind, str(tmp_path), sync_bn), nprocs=2, join=True)
    r = [dict(np.load(tmp_path / f"final{i}.npz")) for i in range(2)]
    assert int(r[0]["step"]) == int(r[1]["step"]) == 2
    assert np.array_equal(r[0]["params"], r[1]["params"])          # one all-reduced gradient, one identical update on every rank
    assert np.array_equal(r[0]["stats"], r[1]["stats"])            # sync BN: identical by construction; otherwise averaged before the save
    ck = tf_bundle.read_checkpoint(str(tmp_path / "TCResNet8Model-2"))      # written by rank 0 only
    assert int(ck["global_step"]) == 2 and np.isfinite(ck["TCResNet8/conv0/weights"]).all()


@pytest.mark.parametrize("sync_bn", [False, True])
def test_train_audio_cli_two_ranks(emu_lib, tmp_path, sync_bn):
    """The reference command line under torch.distributed (2 ranks, gloo): shards of one global batch, replicas stay in lockstep."""
    _cli_two_ranks("emu", tmp_path, sync_bn)


@pytest.mark.gpu
def test_train_audio_cli_two_ranks_hip(hip_lib, tmp_path):
    _cli_two_ranks("hip", tmp_path, True)


BENCH_COLLECTIVES = {"forward": 0, "train": 1.0, "train_tcresnet14_1.5": 1.0, "train_tcresnet14_1.5_3010": 1.0}


def _bench_plain_command(env_extra, batch, legs_ok=True, ranks=2):
    """`python bench.py --gpus N ...` as a PLAIN command (no launcher, WORLD_SIZE unset): bench.py launches its own N ranks under
    torch.distributed.run, rank 0 prints the one JSON line."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--prewarm", "0", "--batch", str(batch),
           "--no-cpu-baseline"]           # (several ranks: the default legs are the training legs, configs[2] and configs[3] at 49 and 98 frames)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]              # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == ranks * batch and out["value"] > 0
    if not legs_ok:
        return out
    assert "secondary_legs" not in out and "dscnn_l_forward" not in out
    # replicas only / ONE all-reduce of the gradient arena per step, for TCResNet8-1.0 and for BASELINE configs[3] (TCResNet14-1.5, global batch 4096 N at full size)
    assert out["collectives_per_step"] == BENCH_COLLECTIVES
    assert out["train"]["value"] > 0 and out["train"]["collectives_per_step"] == 1.0
    for key in ("train_tcresnet14_1.5", "train_tcresnet14_1.5_3010"):      # (98 frames: the reference's own script for this model)
        t14 = out[key]
        assert t14["value"] > 0 and t14["collectives_per_step"] == 1.0 and f"global {ranks * batch}" in t14["workload"] and "all-reduce" in t14["workload"]
    assert "98x40" in out["train_tcresnet14_1.5_3010"]["workload"]
    return out


def test_bench_self_launches_two_ranks(emu_lib):
    """CPU rehearsal (emulator build, gloo): the launcher, rendezvous, barriers, max-over-ranks timing and the JSON contract."""
    out = _bench_plain_command({"TCR_BENCH_EMU": EMU}, 4)
    assert "rehearsal" in out and out["collective_backend"].startswith("gloo")


def test_bench_self_launches_eight_ranks(emu_lib):
    """The driver's `bench.py --gpus 8`, rehearsed on the emulator over gloo: eight ranks' sample offsets, the 1 / global_batch scale of
    the loss gradient, ONE all-reduce per training step for configs[2] and configs[3] -- so that the first run on an 8-GPU node cannot
    fail on arithmetic or control flow (the collective there is RCCL; this is not a measurement)."""
    out = _bench_plain_command({"TCR_BENCH_EMU": EMU, "OMP_NUM_THREADS": "1"}, 2, ranks=8)
    assert "rehearsal" in out and out["collective_backend"].startswith("gloo") and out["config"]["global_batch"] == 16


@pytest.mark.parametrize("how", ["raise", "hang"])
def test_bench_keeps_the_headline_when_a_secondary_leg_fails(emu_lib, how):
    """Several ranks: a leg that raises, or hangs (a collective that never completes) until the watchdog fires, still leaves ONE JSON
    line with the headline and exit code 0."""
    out = _bench_plain_command({"TCR_BENCH_EMU": EMU, "TCR_BENCH_FAIL_LEG": how, "TCR_BENCH_LEG_TIMEOUT": "2"}, 4, legs_ok=False)
    assert "train" not in out and ("injected" if how == "raise" else "not finished") in out["secondary_legs"]["error"]


@pytest.mark.gpu
def test_bench_self_launches_two_ranks_hip(hip_lib):
    """The same plain command on the gfx950 library: two ranks time-sharing the test box's one GPU over gloo (on an N-GPU node the
    identical command without TCR_BENCH_BACKEND runs one rank per GPU over RCCL)."""
    out = _bench_plain_command({"TCR_BENCH_BACKEND": "gloo"}, 256)
    assert "rehearsal" not in out and out["collective_backend"].startswith("gloo")


@pytest.mark.gpu
def test_bench_single_rank_goes_through_rccl(hip_lib):
    """The N > 1 control flow of bench.py (process group with device_id, barriers, max-over-ranks all-reduce, the training leg's
    gradient all-reduce) in a group of ONE rank over backend "nccl" = RCCL: what a 1-GPU box can check of the RCCL path before the
    driver's multi-GPU runs (two ranks cannot share a GPU under RCCL)."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                "TCR_BENCH_FORCE_DIST": "1", "TCR_DP_FORCE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--prewarm", "4", "--batch", "512",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["collective_backend"] == "RCCL (backend nccl)" and out["n_gpus"] == 1 and out["value"] > 0
    assert "secondary_legs" not in out, out["secondary_legs"]
    assert out["train"]["value"] > 0 and out["train_tcresnet14_1.5"]["value"] > 0 and out["train_tcresnet14_1.5_3010"]["value"] > 0
    assert out["collectives_per_step"] == BENCH_COLLECTIVES
