"""Caller-side helpers of the hot path (mirror of /root/reference/src/util.py:46-53)."""
import numpy as np


def softmax_multi_with_log(x, single_values=4, eps=1e-20, temperature=10.0):
    """Action posterior from negated EFE (util.py:46-53,68).  `SM` divides by the temperature, `logSM`
    does not -- kept as the reference computes it.  Host/numpy version for numpy callers; the device
    version is ActiveInferenceModel.action_posterior()."""
    x = np.asarray(x).reshape(-1, single_values)
    x = x - x.max(axis=1, keepdims=True)
    e_x = np.exp(x / temperature)
    tot = e_x.sum(axis=1, keepdims=True)
    return e_x / tot, x - np.log(tot + eps)
