"""CPU ORACLE (test infrastructure, NOT the product): numpy restatement of the occupancy-grid binarisation of
TemporalOccGridEstimator._update (models/occ_grid/temporal_occ_grid.py:389-411) and max_connected_component
(models/utils.py:152-163: exactly res*3 sweeps of a 3^3 max-pool on the label volume, masked by the grid)."""
import numpy as np
from scipy.ndimage import maximum_filter


def connected_component_labels(binaries):
    """max_connected_component (models/utils.py:152-163): every occupied cell ends up with the largest 1-based linear index
    of its 26-connected component, after exactly res[-1] * 3 sweeps."""
    res = binaries.shape
    comp = np.arange(1, binaries.size + 1, dtype=np.int64).reshape(res)
    comp[~binaries] = 0
    for _ in range(res[-1] * 3):
        comp = maximum_filter(comp, size=3, mode="constant", cval=0) * binaries
    return comp


def binarize(occs, res, thre_max, keep_largest_component=True):
    occs = occs.reshape(res).astype(np.float32)
    pooled = maximum_filter(occs, size=3, mode="constant", cval=-np.inf)
    mean = np.float32(pooled[pooled >= 0].astype(np.float64).mean()) if (pooled >= 0).any() else np.float32(np.nan)
    thre = np.float32(min(mean, np.float32(thre_max))) if not np.isnan(mean) else mean
    binaries = pooled > thre
    if keep_largest_component:
        comp = connected_component_labels(binaries)
        labels = comp[binaries]
        if labels.size:
            vals, counts = np.unique(labels, return_counts=True)
            label = vals[np.argmax(counts)]            # smallest label among the most frequent (np.unique sorts)
        else:
            label = 0
        binaries = comp == label
    return binaries, thre
