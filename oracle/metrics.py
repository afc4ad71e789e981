"""ORACLE (test infrastructure, not product): numpy restatement of the reference's ensemble metrics,
`src/utilities/evaluation.py:10-136`.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.

PARITY STATUS: pinned for MSE / SSR (against the imported reference's own functions), DEFINITIONAL for CRPS (the reference
delegates CRPS to xskillscore, which exists neither in this image nor in the reference tree: no output of it can be generated).

Pinning: `ensemble_mse` / `spread_skill_ratio` are checked against the reference's own functions imported from
/root/reference (tests/test_oracle_metrics.py, tests/golden/metrics_*.npz).  `crps_ensemble` lives in third-party
dependencies that are absent here (xskillscore 0.0.24 -> properscoring 0.1 `crps_ensemble`, requirements of the
reference's environment); their published definition is restated -- CRPS of the EMPIRICAL ensemble CDF,
E|X - y| - 0.5 E|X - X'| with 1/N and 1/N^2 weights (no "fair" correction) -- and pinned against a direct numerical
evaluation of int (F_ens(z) - 1[z >= y])^2 dz."""
import numpy as np


def ensemble_mse(predictions: np.ndarray, targets: np.ndarray) -> float:
    """evaluation.py:125-128 with mean_dims=None: MSE of the ensemble mean."""
    return float(np.mean((predictions.mean(axis=0) - targets) ** 2))


def spread_skill_ratio(predictions: np.ndarray, targets: np.ndarray) -> float:
    """evaluation.py:98-116: sqrt(mean population variance over members) / RMSE of the ensemble mean."""
    spread = np.sqrt(np.var(predictions, axis=0).mean())
    return float(spread / np.sqrt(ensemble_mse(predictions, targets)))


def crps_ensemble(predictions: np.ndarray, targets: np.ndarray) -> float:
    """evaluation.py:83-95 -> xs.crps_ensemble(..., dim=all): mean over all points of
    mean_n |x_n - y| - 1/(2 N^2) sum_{n,m} |x_n - x_m|."""
    n = predictions.shape[0]
    x = predictions.reshape(n, -1).astype(np.float64)
    y = targets.reshape(-1).astype(np.float64)
    term1 = np.abs(x - y[None]).mean(axis=0)
    xs = np.sort(x, axis=0)  # sum_{n,m} |x_n - x_m| = 2 sum_i (2 i - N + 1) x_(i)
    w = (2.0 * np.arange(n) - n + 1.0)[:, None]
    term2 = (w * xs).sum(axis=0) / (n * n)
    return float((term1 - term2).mean())


def crps_by_definition(members: np.ndarray, y: float) -> float:
    """int (F(z) - H(z - y))^2 dz for the empirical CDF of `members` (1-D), integrated exactly piecewise."""
    pts = np.sort(np.concatenate([members.astype(np.float64), [float(y)]]))
    total = 0.0
    n = len(members)
    for a, b in zip(pts[:-1], pts[1:]):
        mid = 0.5 * (a + b)
        f = np.count_nonzero(members <= mid) / n
        hv = 1.0 if mid >= y else 0.0
        total += (f - hv) ** 2 * (b - a)
    return total


def evaluate_ensemble_prediction(predictions: np.ndarray, targets: np.ndarray) -> dict:
    """evaluation.py:10-80 (ensemble_dim=0, mean_over_samples=True, no per-member metrics)."""
    assert predictions.shape[1] == targets.shape[0]
    return {"ssr": spread_skill_ratio(predictions, targets), "crps": crps_ensemble(predictions, targets),
            "mse": ensemble_mse(predictions, targets)}
