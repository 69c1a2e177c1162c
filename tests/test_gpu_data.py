"""SURVEY 8(f) rows on the GPU: window gather + label pooling + SpecAugment, sequence packing, StandardScaler fit,
against the numpy oracle (oracle/data_ref.py) and the goldens captured from the imported reference (g7, g8)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sed_crnn_amd import data as d
    return d


def test_window_items_match_reference_dataset(data):
    d = load_golden("g8_window_aug.npz")
    ds = data.HitWindowSet(d["item_mel"], d["item_lab"])
    x, y = ds.gather([int(d["item_start"]), int(d["item_pos_start"])])
    assert x.shape == (2, 1, 40, 64) and y.shape == (2, 8, 1)
    np.testing.assert_array_equal(x[0].cpu().numpy(), d["item_x"])
    np.testing.assert_array_equal(y[0].cpu().numpy(), d["item_y"])
    np.testing.assert_array_equal(x[1].cpu().numpy(), d["item_pos_x"])
    np.testing.assert_array_equal(y[1].cpu().numpy(), d["item_pos_y"])
    # end-of-fold fallback: a start too close to the end is clamped (decorte_datamodule.py:79-82)
    xe, _ = ds.gather([299])
    np.testing.assert_array_equal(xe[0, 0].cpu().numpy(), d["item_mel"][300 - 64:].T)


def test_spec_augment_matches_reference_draws(data):
    from oracle import data_ref
    d = load_golden("g8_window_aug.npz")
    for i in range(3):
        rs = np.random.RandomState(int(d[f"aug_seed{i}"]))
        t, f = data_ref.draw_spec_masks(rs, 40, 64)
        ds = data.HitWindowSet(d[f"aug_in{i}"].T.copy(), np.zeros((64, 1), np.float32))
        x, _ = ds.gather([0], np.asarray([t]), np.asarray([f]))
        np.testing.assert_array_equal(x[0, 0].cpu().numpy(), d[f"aug_out{i}"])


def test_sampler_balance_pooling_and_multichannel(data):
    from oracle import data_ref
    g7 = load_golden("g7_dataset.npz")
    rng = np.random.default_rng(0)
    mel = rng.standard_normal((400, 80)).astype(np.float32)            # 2 channels x 40 mel
    ds = data.HitWindowSet(mel, g7["lab"], n_channels=2, augment=True, seed=3)
    np.testing.assert_array_equal(ds.neg_starts, g7["neg_starts"])
    np.testing.assert_array_equal(ds.pos_frames, g7["pos_frames"])
    assert len(ds) == int(g7["len"])
    idx = np.arange(64)
    starts = ds.draw_starts(idx)
    lab = g7["lab"][:, 0]
    for i, s in zip(idx, starts):
        assert 0 <= s <= 400 - 64
        assert (lab[s:s + 64].max() == 1) == (i % 2 == 0)                # even: holds a positive, odd: clean negative
    t, f = ds.draw_masks(64)
    assert t.min() >= 0 and t.max() < 64 - 8 and f.min() >= 0 and f.max() < 40 - 8
    x, y = ds.gather(starts, t, f)
    for b in (0, 1, 17, 63):
        xr, yr = data_ref.window_item(mel, g7["lab"], int(starts[b]), 64, 8, list(t[b]), list(f[b]), n_channels=2)
        np.testing.assert_array_equal(x[b].cpu().numpy(), xr)
        np.testing.assert_array_equal(y[b].cpu().numpy(), yr)
    loader = data.GpuWindowLoader(ds, batch_size=8, shuffle=True, drop_last=True)
    batches = list(loader)
    assert len(batches) == len(ds) // 8 and batches[0][0].shape == (8, 2, 40, 64) and batches[0][0].is_cuda


def test_pack_sequences_matches_utils(data):
    d = load_golden("g8_window_aug.npz")
    feat = torch.from_numpy(d["pack_feat"]).float().cuda()
    out = data.pack_sequences(feat, 16, n_channels=2, time_last=False)
    np.testing.assert_allclose(out.cpu().numpy(), d["pack_mc"].astype(np.float32), rtol=0, atol=0)
    out_t = data.pack_sequences(feat, 16, n_channels=2, time_last=True)
    np.testing.assert_array_equal(out_t.cpu().numpy(), d["pack_mc"].astype(np.float32).transpose(0, 1, 3, 2))
    assert out.shape == (70 // 16, 2, 16, 40)


@pytest.mark.parametrize("seed", range(10))
def test_pack_sequences_and_window_items_random_shapes_vs_oracle(data, seed):
    """drawn shapes (1-6 channels, 5-128 mel bins, sequence lengths that do and do not divide the fold, windows at the fold's
    edges, with and without SpecAugment masks): `pack_sequences` vs utils.split_in_seqs + split_multi_channels and the
    window kernel vs `HitWindowDataset.__getitem__` + `_pool_labels` + `_spec_augment`, both as restated by the oracle
    (pinned by goldens g7/g8), bit for bit"""
    from oracle import data_ref
    rng = np.random.default_rng(seed)
    C, Fm = int(rng.integers(1, 7)), int(rng.choice([5, 13, 40, 64, 128]))
    S = int(rng.choice([4, 16, 64, 100]))
    N = int(rng.integers(S, 6 * S + 7))
    feat = rng.standard_normal((N, C * Fm)).astype(np.float32)
    want = data_ref.split_multi_channels(data_ref.split_in_seqs(feat, S), C).astype(np.float32)       # [n, C, S, F]
    got = data.pack_sequences(torch.from_numpy(feat).cuda(), S, n_channels=C, time_last=False)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    got_t = data.pack_sequences(torch.from_numpy(feat).cuda(), S, n_channels=C, time_last=True)
    np.testing.assert_array_equal(got_t.cpu().numpy(), want.transpose(0, 1, 3, 2))
    # window items: L a multiple of L_out, starts incl. 0 and the last legal one
    Lo = int(rng.choice([2, 4, 8]))
    L = Lo * int(rng.integers(1, 9))
    n = int(rng.integers(L, 4 * L + 5))
    mel = rng.standard_normal((n, C * Fm)).astype(np.float32)
    lab = (rng.random((n, 1)) > 0.7).astype(np.float32)
    ds = data.HitWindowSet(mel, lab, seq_len_in=L, seq_len_out=Lo, n_channels=C, seed=seed)
    starts = np.array([0, n - L] + list(rng.integers(0, n - L + 1, size=6)), np.int32)
    use_masks = bool(seed % 2) and L > data.TIME_MASK_W and Fm > data.FREQ_MASK_W
    t = f = None
    if use_masks:
        t, f = ds.draw_masks(len(starts))
    x, y = ds.gather(starts, t, f)
    for i, st in enumerate(starts):
        xr, yr = data_ref.window_item(mel, lab, int(st), L, Lo, tmask=None if t is None else t[i], fmask=None if f is None else f[i],
                                      n_channels=C)
        np.testing.assert_array_equal(x[i].cpu().numpy(), np.asarray(xr, np.float32))
        np.testing.assert_array_equal(y[i].cpu().numpy(), np.asarray(yr, np.float32))


def test_window_items_at_the_multichannel_configs_sizes(data):
    """BASELINE configs 3 and 5 window shapes (256 x 2 x 40 and 512 x 4 x 128 = 1 MB per sample): gather + SpecAugment + label
    pooling vs the oracle, bit for bit"""
    from oracle import data_ref
    for C, Fm, L in ((2, 40, 256), (4, 128, 512)):
        rng = np.random.default_rng(C)
        n = 3 * L + 11
        mel = rng.standard_normal((n, C * Fm)).astype(np.float32)
        lab = (rng.random((n, 1)) > 0.9).astype(np.float32)
        ds = data.HitWindowSet(mel, lab, seq_len_in=L, seq_len_out=L // 8, n_channels=C, seed=1)
        starts = np.array([0, n - L, 17, L + 3], np.int32)
        t, f = ds.draw_masks(len(starts))
        x, y = ds.gather(starts, t, f)
        assert x.shape == (4, C, Fm, L) and y.shape == (4, L // 8, 1)
        for i, st in enumerate(starts):
            xr, yr = data_ref.window_item(mel, lab, int(st), L, L // 8, tmask=t[i], fmask=f[i], n_channels=C)
            np.testing.assert_array_equal(x[i].cpu().numpy(), xr)
            np.testing.assert_array_equal(y[i].cpu().numpy(), yr)


@pytest.mark.parametrize("N,Fc", [(1, 3), (2, 40), (1000, 1), (3001, 7), (777, 240), (500, 256), (500, 257), (2000, 512), (300, 768)])
def test_standard_scaler_any_width_vs_oracle(data, N, Fc):
    """feature.py:127-129 on folds of any width (the multichannel configs have 4 x 128 = 512 feature columns): mean_, scale_
    and both transforms against the oracle's restatement of sklearn (pinned by golden g9), incl. constant and huge-mean columns"""
    from oracle import logmel_ref
    rng = np.random.default_rng(N + Fc)
    x = (rng.standard_normal((N, Fc)) * rng.uniform(0.1, 30, Fc) + rng.uniform(-50, 50, Fc)).astype(np.float32)
    if Fc > 2:
        x[:, 1] = 3.25                                            # constant column -> scale 1
        x[:, Fc - 1] += 1e6                                       # huge mean, small spread
    mean, scale = logmel_ref.standardize_fit(x)
    m, sd = data.standard_scaler_fit(torch.from_numpy(x).cuda())
    np.testing.assert_allclose(m.cpu().numpy(), mean, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(sd.cpu().numpy(), scale, rtol=1e-9)
    out = data.standard_scaler_transform(torch.from_numpy(x).cuda(), m, sd)
    np.testing.assert_array_equal(out.cpu().numpy(), logmel_ref.standardize_apply(x, m.cpu().numpy(), sd.cpu().numpy()))


def test_standard_scaler_fit_and_fused_transform(data):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((5000, 40)) * rng.uniform(0.5, 3, 40) + rng.uniform(-5, 5, 40)).astype(np.float32)
    x[:, 7] = 2.5                                                        # constant column: sigma -> 1
    mean, std = data.standard_scaler_fit(torch.from_numpy(x).cuda())
    np.testing.assert_allclose(mean.cpu().numpy(), x.astype(np.float64).mean(0), rtol=1e-12, atol=1e-12)
    ref_sd = x.astype(np.float64).std(0)
    ref_sd[7] = 1.0
    np.testing.assert_allclose(std.cpu().numpy(), ref_sd, rtol=1e-9)


def test_run_epoch_on_gpu_loader(data):
    """the fit loop consumes the device loader unchanged (run_epoch signature of sed.py:128-141)"""
    import sed_crnn_amd as sed
    rng = np.random.default_rng(2)
    mel = rng.standard_normal((600, 40)).astype(np.float32)
    lab = np.zeros((600, 1), np.float32)
    lab[100:110] = 1
    lab[400:420] = 1
    ds = data.HitWindowSet(mel, lab, augment=True, seed=1)
    loader = data.GpuWindowLoader(ds, batch_size=16, shuffle=True, drop_last=True)
    torch.manual_seed(0)
    m = sed.TimePooledCRNN(conv_channels=16, dropout=0.1, gru_hidden=16).cuda()
    opt = sed.FusedAdam(m.parameters(), lr=1e-3)
    l0, p, t = sed.run_epoch(m, loader, sed.BCEWithLogitsLoss(), opt)
    assert p.shape == (len(loader) * 16, 8, 1) and t.shape == p.shape and np.isfinite(l0)
    sc = sed.metrics.compute_scores(p > 0.5, t, frames_in_1_sec=5)
    assert set(sc) == {"f1_overall_1sec", "er_overall_1sec"}


@pytest.mark.parametrize("N,Tp,K,block", [(6, 8, 1, 5), (5, 7, 6, 4), (33, 8, 1, 5), (2, 3, 2, 50), (16, 32, 1, 5)])
def test_device_segment_counts_equal_host_metrics(data, N, Tp, K, block):
    import sed_crnn_amd as sed
    rng = np.random.default_rng(N * 100 + K)
    p = rng.random((N, Tp, K)).astype(np.float32)
    t = (rng.random((N, Tp, K)) > 0.6).astype(np.float32)
    def same(a, b):                                   # bit-equal, NaN == NaN (fewer rows than one block: ER is 0/0)
        return a == b or (np.isnan(a) and np.isnan(b))
    got = sed.metrics.scores_from_counts(sed.metrics.device_counts(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda(), block).cpu().tolist())
    with np.errstate(all="ignore"):
        assert same(got["f1_overall_1sec"], sed.metrics.f1_overall_1sec(p > 0.5, t, block))
        assert same(got["er_overall_1sec"], sed.metrics.er_overall_1sec(p > 0.5, t, block))
        assert same(got["f1_overall_framewise"], sed.metrics.f1_overall_framewise(p > 0.5, t))
        assert same(got["er_overall_framewise"], sed.metrics.er_overall_framewise(p > 0.5, t))
        d = sed.metrics.compute_scores_device(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda(), block)
        h = sed.metrics.compute_scores(p > 0.5, t, block)
    assert all(same(d[k], h[k]) for k in h)
    # and the 17 integers themselves against the oracle's numpy restatement of the reference's intermediate quantities
    from oracle import metrics_ref
    ints = sed.metrics.device_counts(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda(), block).cpu().tolist()
    assert ints == metrics_ref.segment_counts(p > 0.5, t, block)


def test_device_segment_counts_on_a_long_epoch_of_windows(data):
    """an epoch's worth of windows (3 001 x 8 rows, K = 1, block 5: blocks straddle the window boundaries, the last one is
    partial): exact integers vs the oracle"""
    import sed_crnn_amd as sed
    from oracle import metrics_ref
    rng = np.random.default_rng(77)
    p = rng.random((3001, 8, 1)).astype(np.float32)
    t = (rng.random((3001, 8, 1)) > 0.9).astype(np.float32)
    ints = sed.metrics.device_counts(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda(), 5).cpu().tolist()
    assert ints == metrics_ref.segment_counts(p > 0.5, t, 5)


def test_device_counts_golden_and_edges(data):
    import sed_crnn_amd as sed
    d = load_golden("g6_metrics.npz")
    sc = sed.metrics.compute_scores_device(torch.from_numpy(d["p"]).cuda(), torch.from_numpy(d["t"]).cuda(), 5)
    assert sc["f1_overall_1sec"] == float(d["f1_1s"]) and sc["er_overall_1sec"] == float(d["er_1s"])
    z, o = torch.zeros(4, 8, 1).cuda(), torch.ones(4, 8, 1).cuda()
    e = sed.metrics.compute_scores_device(o, z, 5)
    assert e["f1_overall_1sec"] == 0.0 and np.isinf(e["er_overall_1sec"])
    assert np.isnan(sed.metrics.compute_scores_device(z, z, 5)["er_overall_1sec"])
