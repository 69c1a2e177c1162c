"""CPU-side checks of the host logic around the hot path: bench.py's own N-rank launcher (gloo), the fold-pack loader
(feature.py:131-132 / sed.py:115-125), the epoch logic of sed.py:166-202 (best-ER, `no_imp > EARLY_STOP`) with the
device work stubbed out, and the StandardScaler oracle against the sklearn-generated golden g9."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


# ───────────────────────── bench.py launches its own ranks ─────────────────────────
@pytest.mark.timeout(240)
def test_bench_gpus2_starts_two_rank_processes_itself_over_gloo():
    """`python bench.py --gpus 2` with no torchrun environment: the parent only launches; two fresh rank processes
    rendezvous (127.0.0.1), broadcast the parameters, run the staged all-reduce of the flat gradient arena and report the
    world size the process group itself counts.  --plumbing-only leaves the HIP kernels out (no GPU here)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0",
                        "--plumbing-only"], capture_output=True, text=True, env=env, timeout=220)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["metric"] == "plumbing-only" and out["value"] is None      # can never be read as a throughput
    assert out["allreduce_avg_ok"] is True and out["params_in_sync"] is True and out["backend"] == "gloo"


@pytest.mark.timeout(600)
def test_bench_gpus8_plumbing_rendezvous_and_bucket_slicing_at_the_scale_run_size():
    """the driver's SCALE run is N = 8: the launcher, the 127.0.0.1 rendezvous of 8 ranks, the parameter broadcast and the
    staged all-reduce of every arena slice (average over 8 ranks of rank-dependent fills) at that world size, on CPU (gloo)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"                              # 8 rank processes on the build container's 8 CPUs
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "0",
                        "--plumbing-only"], capture_output=True, text=True, env=env, timeout=560)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["parallelism"] == "dp8" and out["backend"] == "gloo"
    assert out["metric"] == "plumbing-only" and out["value"] is None
    assert out["allreduce_avg_ok"] is True and out["params_in_sync"] is True


def test_bench_refuses_a_world_size_that_contradicts_the_flag():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only", "--steps", "1"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "process group has 1 ranks" in (r.stderr + r.stdout)


def test_cpu_baseline_host_description():
    sys.path.insert(0, ROOT)
    import bench
    info = bench.host_cpu_info()
    assert 1 <= info["threads"] <= info["allowed_cpus"] <= info["logical_cpus"]
    assert info["threads"] <= info["physical_cores_allowed"]
    assert isinstance(info["cpu_model"], str)


# ───────────────────────── fold packs ─────────────────────────
def _write_pack(folder, fold, n_train=300, n_val=200, seed=0):
    rng = np.random.default_rng(seed + fold)
    xtr, xva = rng.standard_normal((n_train, 40)).astype(np.float32), rng.standard_normal((n_val, 40)).astype(np.float32)
    ytr, yva = np.zeros((n_train, 1), np.float32), np.zeros((n_val, 1), np.float32)
    ytr[100:106] = 1
    yva[80:84] = 1
    np.savez(os.path.join(folder, f"mbe_mon_fold{fold}.npz"), xtr, ytr, xva, yva)       # positional, like feature.py:131-132
    return xtr, ytr, xva, yva


def test_load_all_npz_reads_the_reference_fold_pack_layout(tmp_path):
    from sed_crnn_amd import data
    written = {f: _write_pack(str(tmp_path), f) for f in range(1, 5)}
    folds = data.load_all_npz(str(tmp_path))
    assert sorted(folds) == [1, 2, 3, 4]
    for f, (xtr, ytr, xva, yva) in written.items():
        fd = folds[f]
        assert list(fd) == ["train_x", "train_y", "val_x", "val_y"]                     # the keys of sed.py:119-123
        np.testing.assert_array_equal(fd["train_x"], xtr)
        np.testing.assert_array_equal(fd["train_y"], ytr)
        np.testing.assert_array_equal(fd["val_x"], xva)
        np.testing.assert_array_equal(fd["val_y"], yva)
    np.testing.assert_array_equal(data.load_fold_npz(str(tmp_path), 3)["val_x"], written[3][2])
    with pytest.raises(FileNotFoundError):
        data.load_fold_npz(str(tmp_path), 5)
    np.savez(tmp_path / "mbe_mon_fold5.npz", a=np.zeros(3))                             # named arrays: not a fold pack
    with pytest.raises(ValueError, match="not a fold pack"):
        data.load_fold_npz(str(tmp_path), 5)
    np.savez(tmp_path / "mbe_mon_fold6.npz", np.zeros((5, 40)), np.zeros((4, 1)), np.zeros((2, 40)), np.zeros((2, 1)))
    with pytest.raises(ValueError, match="frames"):
        data.load_fold_npz(str(tmp_path), 6)


# ───────────────────────── epoch logic of sed.py:166-202 (device work stubbed) ─────────────────────────
class _FakeTally:
    def __init__(self, loss, er, f1=0.5):
        self._loss, self._er, self._f1 = loss, er, f1

    def mean_loss(self):
        return self._loss

    def scores(self, fps):
        assert fps == 5
        return {"f1_overall_1sec": self._f1, "er_overall_1sec": self._er}


def _run_fit(monkeypatch, val_ers, early_stop, max_epochs=50, tmp=None):
    import importlib
    fitmod = importlib.import_module("sed_crnn_amd.fit")       # (the package re-exports the function `fit` under that name)
    calls = {"n": 0, "saved": []}

    def fake_epoch(model, loader, loss_fn, optim=None, device=None):
        if optim is None:                                   # validation pass of this epoch
            er = val_ers[calls["n"]]
            calls["n"] += 1
            return _FakeTally(0.3, er)
        return _FakeTally(0.7, 1.0)
    monkeypatch.setattr(fitmod, "run_epoch_device", fake_epoch)
    model = torch.nn.Linear(2, 1)
    real_save = torch.save

    def spy_save(obj, path):
        calls["saved"].append((calls["n"], path))
        real_save(obj, path)
    monkeypatch.setattr(torch, "save", spy_save)
    res = fitmod.fit(model, None, None, None, optim=object(), max_epochs=max_epochs, early_stop=early_stop,
                     save_best=str(tmp / "best_fold1.pt") if tmp else None)
    return res, calls


def test_fit_keeps_best_er_and_stops_after_early_stop_plus_one_stale_epochs(monkeypatch, tmp_path):
    # improvement at epochs 1, 2 and 4; ties do not count (strict <); then nothing better
    ers = [0.9, 0.8, 0.8, 0.5, 0.5, 0.6, 0.7, 0.5, 0.9, 0.9, 0.9]
    res, calls = _run_fit(monkeypatch, ers, early_stop=3, tmp=tmp_path)
    # no_imp after epoch 4 is 0; epochs 5,6,7,8 make it 1,2,3,4 -> `no_imp > 3` fires at epoch 8 (sed.py:200-202)
    assert res["best_er"] == 0.5 and res["best_epoch"] == 4
    assert len(res["history"]) == 8 and calls["n"] == 8
    assert [e for e, _ in calls["saved"]] == [1, 2, 4]                     # a checkpoint at every improvement only
    sd = torch.load(tmp_path / "best_fold1.pt", weights_only=True)          # a bare state_dict (sed.py:198-199)
    assert set(sd) == {"weight", "bias"}
    rec = res["history"][3]
    assert rec == dict(epoch=4, train_loss=0.7, val_loss=0.3, train_f1=0.5, val_f1=0.5, val_er=0.5)


def test_fit_runs_to_max_epochs_without_early_stop_and_handles_nan_er(monkeypatch):
    res, calls = _run_fit(monkeypatch, [float("nan")] * 5 + [0.4], early_stop=40, max_epochs=6)
    assert calls["n"] == 6 and res["best_epoch"] == 6 and res["best_er"] == 0.4      # nan < inf is False: never "best"
    res, calls = _run_fit(monkeypatch, [float("inf")] * 4, early_stop=1, max_epochs=10)
    assert calls["n"] == 2 and res["best_epoch"] == 0 and res["best_er"] == float("inf")   # 2 stale epochs: `no_imp > 1`


# ───────────────────────── StandardScaler oracle vs sklearn (g9) ─────────────────────────
def test_scaler_oracle_reproduces_the_sklearn_golden():
    """feature.py:127-129: fit_transform on the train split, transform on the test split.  The golden was written by
    scikit-learn itself (oracle/make_goldens.py g9); the numpy restatement must reproduce mean_, scale_ (incl. the
    constant / near-constant / huge-mean columns) and the float32 transforms."""
    from oracle import logmel_ref
    d = load_golden("g9_scaler.npz")
    xtr, xte = logmel_ref.scaler_fixture_inputs(int(d["seed"]))
    assert xtr.shape[0] == int(d["n_samples_seen_"])
    mean, scale = logmel_ref.standardize_fit(xtr)
    np.testing.assert_allclose(mean, d["mean_"], rtol=1e-14, atol=1e-14)
    np.testing.assert_allclose(scale, d["scale_"], rtol=1e-12)
    assert scale[3] == 1.0 and scale[4] == 1.0                       # exactly constant columns (3.25 and 0)
    assert 0 < scale[5] < 1e-9                                       # constant up to ONE float32 ulp: sklearn keeps sigma
    np.testing.assert_array_equal(logmel_ref.standardize_apply(xtr, mean, scale), d["train_t"])
    np.testing.assert_array_equal(logmel_ref.standardize_apply(xte, mean, scale), d["test_t"])


# ───────────────────────── feature.py's per-recording cache and label raster ─────────────────────────
def test_label_raster_and_per_recording_cache_format(tmp_path):
    from oracle import data_ref
    from sed_crnn_amd import feature
    hits = [(0.0, 0.05), (1.0, 1.2), (2.32, 2.33), (9.9, 10.5)]
    n = 440
    lbl = feature.rasterize_hits(n, hits)
    np.testing.assert_array_equal(lbl, data_ref.rasterize_hits_ref(n, hits))
    assert lbl.shape == (n, 1) and lbl.dtype == np.float32
    assert lbl[:3, 0].tolist() == [1, 1, 1] and lbl[3, 0] == 0                     # [floor(0), ceil(0.05*43.07)=3)
    assert lbl[43, 0] == 1 and lbl[42, 0] == 0 and lbl[51, 0] == 1 and lbl[52, 0] == 0   # 1.0 s .. 1.2 s
    assert lbl[426:, 0].all() and lbl.sum() == 3 + 9 + 2 + 14                      # the last hit is clipped at the end
    mbe = np.random.default_rng(0).standard_normal((n, 40)).astype(np.float32)
    path = str(tmp_path / "rec01_mon.npz")
    feature.save_video_npz(path, mbe, lbl)
    with np.load(path) as d:
        assert d.files == ["arr_0", "arr_1"]                                        # what feature.py:75-76 reads back
    a, b = feature.load_video_npz(path)
    np.testing.assert_array_equal(a, mbe)
    np.testing.assert_array_equal(b, lbl)
    np.savez(tmp_path / "bad_mon.npz", mbe, lbl[:-1])
    with pytest.raises(ValueError, match="label frames"):
        feature.load_video_npz(str(tmp_path / "bad_mon.npz"))


# ───────────────────────── epoch scores from integer counts (host formulas, no GPU) ─────────────────────────
@pytest.mark.parametrize("n,tp,k,block,seed", [(6, 8, 1, 5, 1234), (5, 7, 6, 4, 1), (33, 8, 1, 5, 2), (2, 3, 2, 50, 3), (16, 32, 1, 5, 4),
                                                (1, 4, 1, 5, 5), (7, 8, 3, 8, 6)])
def test_scores_from_counts_equal_the_reference_metrics(n, tp, k, block, seed):
    """`metrics.scores_from_counts` applies the reference's float64 formulas (metrics.py:25-29,43-44) to the 17 integers
    the device kernel returns; here the integers come from the oracle's numpy restatement, so the host half of the
    device-metrics path is pinned without a GPU: all four scores and the confusion matrix, incl. a partial last block
    (F1 keeps it, ER drops it), fewer rows than one block, and K = 6."""
    from oracle import metrics_ref
    from sed_crnn_amd import metrics
    rng = np.random.default_rng(seed)
    p = rng.random((n, tp, k)).astype(np.float32)
    t = (rng.random((n, tp, k)) > 0.6).astype(np.float32)
    o = p > 0.5
    counts = metrics_ref.segment_counts(o, t, block)
    assert len(counts) == metrics.N_COUNTS
    got = metrics.scores_from_counts(counts)
    with np.errstate(divide="ignore", invalid="ignore"):
        want = {"f1_overall_framewise": metrics_ref.f1_framewise(o, t), "er_overall_framewise": metrics_ref.er_framewise(o, t),
                "f1_overall_1sec": metrics_ref.f1_1sec(o, t, block), "er_overall_1sec": metrics_ref.er_1sec(o, t, block)}
    for key, w in want.items():
        g = got[key]
        assert (np.isnan(w) and np.isnan(g)) or g == w, (key, g, w)          # bit-equal float64 (nan when Nref = 0 blocks)
    o2, t2 = o.reshape(-1, k), t.reshape(-1, k)
    assert got["cm"].tolist() == [[int(((t2 == 0) & ~o2).sum()), int(((t2 == 0) & o2).sum())],
                                  [int(((t2 == 1) & ~o2).sum()), int(((t2 == 1) & o2).sum())]]
    # the host metrics module itself (the reference's API names) agrees as well
    assert metrics.compute_scores(o, t, block) == {"f1_overall_1sec": want["f1_overall_1sec"], "er_overall_1sec": want["er_overall_1sec"]} \
        or np.isnan(want["er_overall_1sec"])


def test_known_answer_of_the_reference_metrics_through_the_counts():
    """rng(1234), p > 0.5 vs t, block 5 -> f1 = 0.8888888888888887, er = 0.125 (48 rows: 10 ceil / 9 floor blocks)."""
    from oracle import metrics_ref
    from sed_crnn_amd import metrics
    rng = np.random.default_rng(1234)
    p = rng.random((6, 8, 1)).astype(np.float32)
    t = (rng.random((6, 8, 1)) > 0.6).astype(np.float32)
    s = metrics.scores_from_counts(metrics_ref.segment_counts(p > 0.5, t, 5))
    assert s["f1_overall_1sec"] == 0.8888888888888887 and s["er_overall_1sec"] == 0.125
    assert s["f1_overall_framewise"] == 0.34146341463414626 and s["er_overall_framewise"] == 1.588235294117647


# ───────────────────────── window sampler: host-side draws (sed.py:55-79, decorte_datamodule.py:39-49) ─────────────────────────
def test_window_sampler_draws_balanced_valid_windows_and_masks():
    """even dataset index -> a window that contains a positive frame, odd -> a window with none (sed.py:64-70); starts stay
    inside the fold; SpecAugment offsets follow `np.random.randint(0, n - W)` (exclusive upper bound); same seed, same draws."""
    from sed_crnn_amd import data
    rng = np.random.default_rng(0)
    n = 5000
    lab = np.zeros((n, 1), np.float32)
    for s0 in rng.integers(0, n - 10, size=12):
        lab[s0:s0 + rng.integers(1, 9)] = 1
    lab[0] = 1                                             # a positive at the very start and the very end of the fold
    lab[-1] = 1
    mel = rng.standard_normal((n, 40)).astype(np.float32)
    ds = data.HitWindowSet(mel, lab, device="cpu", seed=5)
    assert len(ds) == 2 * int(lab.sum())                   # sed.py:62
    idx = np.arange(len(ds))
    st = ds.draw_starts(idx)
    L = data.SEQ_LEN_IN
    assert st.min() >= 0 and st.max() <= n - L
    pos = lab[:, 0] == 1
    cs = np.concatenate([[0], np.cumsum(pos)])
    npos = cs[st + L] - cs[st]
    assert (npos[idx % 2 == 0] >= 1).all() and (npos[idx % 2 == 1] == 0).all()
    t, f = ds.draw_masks(64)
    assert t.shape == f.shape == (64, data.MASKS_PER_EX)
    assert t.min() >= 0 and t.max() < L - data.TIME_MASK_W and f.min() >= 0 and f.max() < 40 - data.FREQ_MASK_W
    ds2 = data.HitWindowSet(mel, lab, device="cpu", seed=5)
    np.testing.assert_array_equal(ds2.draw_starts(idx), st)
    # the clean-negative starts are exactly the windows without a positive frame
    neg = data.find_clean_negatives(lab, L)
    assert set(neg.tolist()) == {s1 for s1 in range(n - L + 1) if cs[s1 + L] - cs[s1] == 0}
