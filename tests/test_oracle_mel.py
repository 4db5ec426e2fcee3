"""Known-answer tests for the oracle's log-mel front end.  torchaudio (where the reference's arithmetic lives) is absent
from /root/reference and from this image, so this part of the oracle is PARITY UNPINNED; it is pinned here only by
(a) an independent numpy restatement of the STFT (rfft of reflect-padded, Hann-windowed frames) and of the filterbank
formulas, and (b) the self-consistency probes recorded in SURVEY.md Appendix A.1 (sine -> mel bin 21, 42.16 dB)."""
import numpy as np
import pytest
import torch

from oracle import atst_oracle as O


def numpy_log_mel(wave, win=1024):
    x = np.pad(wave.astype(np.float64), (512, 512), mode="reflect")
    T = 1 + len(wave) // 160
    w = np.zeros(1024)
    n = np.arange(win)
    left = (1024 - win) // 2
    w[left:left + win] = 0.5 - 0.5 * np.cos(2 * np.pi * n / win)              # periodic Hann, centred zero padding
    frames = np.stack([x[t * 160:t * 160 + 1024] * w for t in range(T)])
    power = np.abs(np.fft.rfft(frames, axis=1)) ** 2                            # [T, 513]
    freqs = np.linspace(0, 8000, 513)
    m_pts = np.linspace(O.hz_to_mel_htk(60.0), O.hz_to_mel_htk(7800.0), 66)
    f_pts = O.mel_to_hz_htk(m_pts)
    fb = np.zeros((513, 64))
    for m in range(64):
        lo, c, hi = f_pts[m], f_pts[m + 1], f_pts[m + 2]
        fb[:, m] = np.maximum(0.0, np.minimum((freqs - lo) / (c - lo), (hi - freqs) / (hi - c)))
    mel = power @ fb                                                            # [T, 64]
    db = 10 * np.log10(np.maximum(mel, 1e-10))
    db = np.maximum(db, db.max() - 80.0)
    return ((db - O.DB_MIN) / (O.DB_MAX - O.DB_MIN) * 2 - 1).T


def test_filterbank_shape_and_support():
    fb = O.mel_filterbank()
    assert fb.shape == (513, 64)
    assert int((fb.sum(1) > 0).sum()) == 496                                    # SURVEY Appendix A.1 probe
    assert float(fb.min()) >= 0.0 and float(fb.max()) <= 1.0 + 1e-6


def test_log_mel_vs_numpy_restatement():
    for win in (1024, 640):
        wave = O.recipe_wave(1, 32000, seed=3)[0]
        got = O.log_mel(wave[None], win_length=win)[0, 0].numpy()
        want = numpy_log_mel(wave.numpy(), win)
        assert got.shape == want.shape == (64, 201)
        assert float(np.abs(got - want).max()) < 2e-4


def test_sine_and_silence_known_answers():
    t = torch.arange(160000) / 16000.0
    m = O.log_mel((0.5 * torch.sin(2 * np.pi * 1000.0 * t))[None])[0, 0]
    db = (m + 1) / 2 * (O.DB_MAX - O.DB_MIN) + O.DB_MIN
    assert m.shape == (64, 1001)
    assert int(db.mean(1).argmax()) == 21 and abs(float(db.max()) - 42.16) < 0.02
    assert abs(float(db.min()) - (float(db.max()) - 80.0)) < 1e-4
    m640 = O.log_mel((0.5 * torch.sin(2 * np.pi * 1000.0 * t))[None], win_length=640)[0, 0]
    assert abs(float(((m640 + 1) / 2 * (O.DB_MAX - O.DB_MIN) + O.DB_MIN).max()) - 40.02) < 0.02
    sil = O.log_mel(torch.zeros(1, 16000))
    assert torch.allclose(sil, torch.full_like(sil, (-100.0 - O.DB_MIN) / (O.DB_MAX - O.DB_MIN) * 2 - 1))
    noise = O.log_mel(O.recipe_wave(1, 160000, seed=1234))
    assert abs(float(noise.mean()) - 0.42) < 0.03                               # N(0,0.1^2) -> about +0.42 after MinMax


def test_block_mask_properties():
    rs = np.random.RandomState(0)
    for _ in range(50):
        m = O.block_mask(250, 0.65, 5, rng=rs)
        assert m.dtype == bool and m.shape == (250,)
        runs = np.diff(np.flatnonzero(np.diff(np.concatenate(([0], m.astype(int), [0])))))[::2]
        assert runs.min() >= 5 or m[-5:].any()                                  # spans of length 5 (may merge / clip at the end)
        assert 0.35 < m.mean() < 0.66                                           # overlap => realised ratio below 0.65


def test_log_mel_vs_third_party_audio_utils():
    """A THIRD-PARTY cross-check that is present in the image: transformers.audio_utils (Hugging Face's numpy port of the torchaudio /
    librosa front ends -- `mel_filter_bank(norm=None, mel_scale="htk")` documents itself as torchaudio.functional.melscale_fbanks,
    `spectrogram(center=True, pad_mode="reflect", power=2)` as torchaudio's Spectrogram, `power_to_db(db_range)` as AmplitudeToDB(top_db)).
    It is not torchaudio, so the row stays "parity unpinned" in the strict sense, but it is code this repo did not
    write: filterbank, STFT power and dB stage of the oracle agree with it, for both recipes' windows and for the 32 kHz / 128-band geometry."""
    A = pytest.importorskip("transformers.audio_utils")
    for sr, n_mels, win in ((16000, 64, 1024), (16000, 64, 640), (32000, 128, 1024)):
        fb_hf = A.mel_filter_bank(num_frequency_bins=513, num_mel_filters=n_mels, min_frequency=60.0, max_frequency=7800.0, sampling_rate=sr,
                                  norm=None, mel_scale="htk")
        fb = O.mel_filterbank(n_mels=n_mels, sample_rate=sr).numpy()
        assert fb_hf.shape == fb.shape and np.abs(fb_hf - fb).max() < 2e-5, (sr, n_mels, np.abs(fb_hf - fb).max())   # fp32 (torchaudio op order) vs float64: 4e-6 / 1e-5
        rng = np.random.default_rng(sr + win)
        t = np.arange(sr) / sr
        wave = (0.1 * rng.standard_normal(sr) + 0.3 * np.sin(2 * np.pi * 440.0 * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 3.0 * t))).astype(np.float32)
        n = np.arange(win)
        window = np.zeros(1024); left = (1024 - win) // 2
        window[left:left + win] = 0.5 - 0.5 * np.cos(2 * np.pi * n / win)                # periodic Hann(win), centred in the 1024-point frame (torch.stft)
        spec = A.spectrogram(wave.astype(np.float64), window, frame_length=1024, hop_length=160, fft_length=1024, power=2.0, center=True,
                             pad_mode="reflect", mel_filters=fb_hf, mel_floor=1e-10, dtype=np.float64)       # [n_mels, T]
        db = A.power_to_db(spec, reference=1.0, min_value=1e-10, db_range=80.0)
        want = (db - O.DB_MIN) / (O.DB_MAX - O.DB_MIN) * 2 - 1
        got = O.log_mel(torch.from_numpy(wave)[None], win, n_mels=n_mels, sample_rate=sr)[0, 0].numpy()
        assert got.shape == want.shape, (got.shape, want.shape)
        assert np.abs(got - want).max() < 2e-4, (sr, n_mels, win, np.abs(got - want).max())
