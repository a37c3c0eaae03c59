"""Weight sets that stress the 16-bit plans' scales: the trained ETH-CNN blobs are absent from the reference, every
other fast-plan test uses the seeded generator -- a real checkpoint may have a few outlier weights or heavy tails, which make the
GUARANTEED activation bounds (and with them the fp16 scales) loose.  Built from the seeded blob, tensor by tensor (oracle.tensor_views)."""
import numpy as np


def _tensors(oracle, blob):
    out = oracle.tensor_views(blob)  # name -> view into blob
    return out


def outlier_per_tensor(oracle, seed, gain, factor):
    """one weight of EVERY tensor multiplied by `factor` (the largest-magnitude one: the bound of the layer grows by about that)"""
    blob = oracle.synth_blob(seed, gain).copy()
    for name, v in _tensors(oracle, blob).items():
        flat = v.reshape(-1)
        k = int(np.argmax(np.abs(flat)))
        flat[k] *= factor
    return blob


def outlier_in(oracle, seed, gain, factor, which):
    """one outlier in the tensors whose name contains one of `which` only"""
    blob = oracle.synth_blob(seed, gain).copy()
    for name, v in _tensors(oracle, blob).items():
        if any(w in name for w in which):
            flat = v.reshape(-1)
            flat[int(np.argmax(np.abs(flat)))] *= factor
    return blob


def heavy_tailed(oracle, seed, gain, df=2.0):
    """every weight matrix redrawn with Student-t tails (df = 2: infinite variance) at the seeded blob's median magnitude"""
    blob = oracle.synth_blob(seed, gain).copy()
    rng = np.random.default_rng(1000 + seed)
    for name, v in _tensors(oracle, blob).items():
        if v.ndim < 2:
            continue
        med = float(np.median(np.abs(v)))
        t = rng.standard_t(df, size=v.shape).astype(np.float32)
        t *= med / max(float(np.median(np.abs(t))), 1e-20)
        v[...] = t
    return blob


def cancelling_pairs(oracle, seed, gain, mag):
    """two taps of output channel 0 of every conv kernel set to +mag and -mag: the response to FLAT input is unchanged (the pair cancels
    exactly), the guaranteed bounds -- sums of |w| -- grow by mag per layer, i.e. by mag^3 for the features.  On flat / smooth content
    the activations stay small while the fp16 scales are sized for 'mag^3': the loose-bound case at its purest, without saturating
    every output (an outlier weight alone drives the sigmoids to 0 / 1, where no plan can differ from another)."""
    blob = oracle.synth_blob(seed, gain).copy()
    for name, v in _tensors(oracle, blob).items():
        if v.ndim == 4:  # conv kernels [ky][kx][ci][co]
            v[0, 0, 0, 0] = mag
            v[0, 1, 0, 0] = -mag
    return blob
