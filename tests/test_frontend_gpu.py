"""HIP log-mel front end vs the CPU oracle (torch.stft based restatement of torchaudio's transform; parity unpinned at
the torchaudio boundary, see oracle header).  Tolerance: |d| <= 1e-3 on the min-max normalised output (= 0.065 dB),
median error <= 2e-5; the known-answer checks mirror tests/test_oracle_mel.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from audiossl_amd.frontend import LogMelFrontend  # noqa: E402
from oracle import atst_oracle as O  # noqa: E402


@pytest.mark.parametrize("win,n", [(1024, 160000), (640, 160000), (1024, 16000), (1024, 96123)])
def test_mel_vs_oracle(win, n):
    wave = O.recipe_wave(3, n, seed=11)
    wave[1] *= 0.01                                    # a quiet clip: exercises the per-clip max / -80 dB floor
    wave[2, n // 2:] = 0.0                             # half silence: many bins at the floor
    got = LogMelFrontend(win)(wave.cuda()).cpu()
    want = O.log_mel(wave, win_length=win)
    assert got.shape == want.shape == (3, 1, 64, 1 + n // 160)
    d = (got - want).abs()
    assert float(d.max()) < 1e-3, float(d.max())
    assert float(d.median()) < 2e-5


def test_mel_known_answers():
    fe = LogMelFrontend(1024)
    t = torch.arange(160000) / 16000.0
    sine = (0.5 * torch.sin(2 * np.pi * 1000.0 * t))[None]
    m = fe(sine.cuda()).cpu()[0, 0]
    db = (m + 1) / 2 * (O.DB_MAX - O.DB_MIN) + O.DB_MIN
    assert int(db.mean(1).argmax()) == 21                              # 1 kHz lands in mel bin 21
    assert abs(float(db.max()) - 42.16) < 0.05
    assert abs(float(db.min()) - (float(db.max()) - 80.0)) < 1e-3      # top_db clamp, one max per clip
    sil = fe(torch.zeros(1, 160000).cuda()).cpu()
    assert torch.allclose(sil, torch.full_like(sil, (-100.0 - O.DB_MIN) / (O.DB_MAX - O.DB_MIN) * 2 - 1), atol=1e-6)


@pytest.mark.parametrize("win,n", [(1024, 320000), (640, 64000)])
def test_mel_hires_vs_oracle(win, n):
    """The reference's `sr` / `n_mels` transform parameters (methods/atstframe/transform.py:14-16) at 32 kHz / 128 bands
    (BASELINE.json configs[4]); the restated oracle is parity-unpinned at the torchaudio boundary like the 64-band one."""
    wave = O.recipe_wave(3, n, seed=13)
    wave[1] *= 0.01
    wave[2, n // 3:] = 0.0
    got = LogMelFrontend(win, sr=32000, n_mels=128)(wave.cuda()).cpu()
    want = O.log_mel(wave, win_length=win, n_mels=128, sample_rate=32000)
    assert got.shape == want.shape == (3, 1, 128, 1 + n // 160)
    d = (got - want).abs()
    assert float(d.max()) < 1e-3, float(d.max())
    assert float(d.median()) < 2e-5


def test_mel_strided_rows_and_shared_output_buffer():
    """Views are slices of a longer waveform buffer (row stride > view length) and land in ONE [V*B, 1, 64, T] buffer: same
    numbers as contiguous inputs / separate outputs, no copies (what bench.py and the training transform feed the engine)."""
    fe = LogMelFrontend(1024)
    buf = O.recipe_wave(4, 192000, seed=17).cuda()
    offs = (1001, 20003)                                  # odd offsets: rows are not 8-byte aligned
    group = torch.empty(8, 1, 64, 1001, device="cuda")
    for v, o in enumerate(offs):
        fe(buf[:, o:o + 160000], out=group[4 * v:4 * v + 4])
    for v, o in enumerate(offs):
        ref = fe(buf[:, o:o + 160000].contiguous())
        assert torch.equal(group[4 * v:4 * v + 4], ref)
