"""Input stage: PCM decode + crop/pad + shift + background mix.  Bit-exact against oracle/augment_ref.py."""
import os
import struct

import numpy as np
import pytest
import torch

import tcresnet_amd as T
from oracle import augment_ref as A
from tests import common as Cm


def _pools(rng, n_clips=12, n_bg=3, desired=16000):
    lens = [desired, max(desired - 3000, 2), desired + 5000, 0, 1, desired - 1] + [int(x) for x in rng.randint(desired // 2, desired * 5 // 4, n_clips - 6)]
    clips = [rng.randint(-32768, 32768, n).astype(np.int16) for n in lens]
    clips[0][:4] = [-32768, 32767, 0, -1]            # extremes: -1.0 exactly, just below +1.0
    bgs = [rng.randint(-32768, 32768, int(n)).astype(np.int16) for n in rng.randint(desired, 4 * desired, n_bg)]
    return clips, bgs


def _run(lib, clips, bgs, idx, shift, bg_idx, bg_crop, vol, desired, fn_name):
    from tcresnet_amd.datasets import augmentation_factory as F
    from tcresnet_amd import runtime
    runtime.set_default(lib, Cm.device_of(lib))
    try:
        pool, bg = F.PcmPool(clips), F.PcmPool(bgs)
        fn = F.get_audio_augmentation_fn(fn_name)
        out = fn(pool, idx, desired, "wav", 16000, background_data=bg, is_training=True, draws=(shift, bg_idx, bg_crop, vol))
        return out.cpu().numpy()[..., 0], pool, bg
    finally:
        runtime.set_default(None, None)


def _check(lib, batch, desired=16000, seed=0):
    rng = np.random.RandomState(seed)
    clips, bgs = _pools(rng, desired=desired)
    idx = rng.randint(0, len(clips), batch)
    idx[:6] = np.arange(6)
    shift = rng.randint(-desired // 10, desired // 10, batch).astype(np.int32)
    shift[:4] = [0, -desired // 10, desired // 10 - 1, 7]
    bg_idx = rng.randint(0, len(bgs), batch)
    bg_crop = np.array([rng.randint(0, len(bgs[i]) - desired + 1) for i in bg_idx], dtype=np.int64)
    vol = np.where(rng.uniform(size=batch) < 0.8, rng.uniform(0, 0.1, batch), 0.0).astype(np.float32)
    vol[0], vol[1] = 1.0, 0.0                        # forces clipping on the full-scale clip; un-mixed element
    got, pool, bg = _run(lib, clips, bgs, idx, shift, bg_idx, bg_crop, vol, desired, "anchored_slice_or_pad_with_shift")
    pcm_pool, bg_pool = pool.data.cpu().numpy(), bg.data.cpu().numpy()
    ref = A.augment_batch(pcm_pool, pool.offsets[idx], pool.lengths[idx], shift, bg_pool, bg.offsets[bg_idx] + bg_crop, vol, desired)
    assert got.dtype == np.float32 and np.array_equal(got, ref)          # bit exact
    assert np.abs(got).max() <= 1.0 and (np.abs(got) == 1.0).any()
    # anchored_slice_or_pad ignores the shift; no_augmentation_audio also ignores the background
    got2, _, _ = _run(lib, clips, bgs, idx, shift, bg_idx, bg_crop, vol, desired, "anchored_slice_or_pad")
    assert np.array_equal(got2, A.augment_batch(pcm_pool, pool.offsets[idx], pool.lengths[idx], np.zeros(batch, np.int32), bg_pool,
                                                bg.offsets[bg_idx] + bg_crop, vol, desired))
    got3, _, _ = _run(lib, clips, bgs, idx, shift, bg_idx, bg_crop, vol, desired, "no_augmentation_audio")
    assert np.array_equal(got3, np.stack([A.decode_wav(clips[i], desired) for i in idx]))
    return got


def test_oracle_semantics():
    """decode / shift / mix follow the reference graph (augmentation_factory.py:104-155, 92-97)."""
    a = A.decode_wav(np.array([-32768, 16384, 1], np.int16), 5)
    assert np.array_equal(a, np.array([-1.0, 0.5, 1.0 / 32768, 0, 0], np.float32))
    assert np.array_equal(A.decode_wav(np.arange(10, dtype=np.int16), 4), np.arange(4, dtype=np.float32) / 32768)
    x = np.arange(1, 7, dtype=np.float32)
    assert np.array_equal(A.shift_audio(x, 2), [0, 0, 1, 2, 3, 4])          # pad front, keep the first N
    assert np.array_equal(A.shift_audio(x, -2), [3, 4, 5, 6, 0, 0])         # pad back, drop the first |s|
    assert np.array_equal(A.shift_audio(x, 0), x)
    m = A.mix_background(np.array([0.9, -0.9, 0.1], np.float32), np.array([1.0, -1.0, 0.5], np.float32), 0.5)
    assert np.array_equal(m, np.array([1.0, -1.0, np.float32(0.25) + np.float32(0.1)], np.float32))
    rng = np.random.RandomState(3)
    for _ in range(100):
        s, bi, bc, v = A.draw(rng, 16000, 3, [16000, 40000, 20000], True, 0.8, 0.1, True)
        assert -1600 <= s < 1600 and 0 <= bi < 3 and 0 <= bc <= [16000, 40000, 20000][bi] - 16000 and 0.0 <= v <= 0.1
    assert A.draw(rng, 16000, 3, [16000] * 3, False, 0.8, 0.1, True)[3] == 0.0     # evaluation: background volume 0 (:76-77)


def test_wav_reader(tmp_path):
    from tcresnet_amd.datasets.augmentation_factory import read_wav_pcm16
    pcm = np.array([0, 1, -1, 32767, -32768, 1234], dtype="<i2")
    body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"LIST" + struct.pack("<I", 4) + b"abcd" \
        + b"data" + struct.pack("<I", pcm.nbytes) + pcm.tobytes()
    path = tmp_path / "a.wav"
    path.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    assert np.array_equal(read_wav_pcm16(str(path)), pcm)
    path.write_bytes(b"RIFX" + b"\0" * 40)
    with pytest.raises(ValueError):
        read_wav_pcm16(str(path))


def test_augment_emulator(emu_lib):
    _check(emu_lib, batch=24)
    _check(emu_lib, batch=7, desired=1003, seed=1)     # desired_samples not a multiple of 4


def test_augment_draw_order_matches_oracle(emu_lib):
    """Without explicit draws the host mirror consumes its generator exactly like oracle.draw (the reference's per-element order)."""
    from tcresnet_amd.datasets import augmentation_factory as F
    from tcresnet_amd import runtime
    rng = np.random.RandomState(5)
    clips, bgs = _pools(rng)
    runtime.set_default(emu_lib, "cpu")
    try:
        pool, bg = F.PcmPool(clips), F.PcmPool(bgs)
        F.anchored_slice_or_pad_with_shift(pool, list(range(8)), 16000, background_data=bg, is_training=True, rng=np.random.RandomState(11))
        got = F.anchored_slice_or_pad_with_shift.last_draws
    finally:
        runtime.set_default(None, None)
    r2 = np.random.RandomState(11)
    want = [A.draw(r2, 16000, len(bgs), [len(b) for b in bgs], True, 0.8, 0.1, True) for _ in range(8)]
    for i, (s, bi, bc, v) in enumerate(want):
        assert (got[0][i], got[1][i], got[2][i]) == (s, bi, bc) and got[3][i] == np.float32(v)


@pytest.mark.gpu
def test_augment_gpu_full_batch(hip_lib):
    """Batch 4096 x 16000 samples, bit exact; then straight into the front-end (the kernel's output is its input)."""
    got = _check(hip_lib, batch=4096)
    fe = Cm.make_frontend(hip_lib, 640, 320)
    feat = fe(torch.from_numpy(got[:64]).cuda())
    from oracle import numpy_ref as R
    ref = R.mfcc(got[:64].astype(np.float64), R.FRONTEND_4020)
    assert np.abs(fe.reference_view(feat)[..., 0].cpu().numpy() - ref).max() < Cm.MFCC_TOL
