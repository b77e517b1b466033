"""colorlut on RGB10A2_LE (the third format of d3d12colorlut's caps, d3d12colorlut/imp.rs:236-244): the oracle's extension
of the CPU element's arithmetic to 10-bit samples (v / 1023, ..., round(clamp * 1023), alpha bits copied) against the
numpy twin and against properties.  Self-defined extension: parity unpinned by construction (the D3D12 shader is not
bit-defined)."""
import numpy as np
import pytest

from tests import cubes, np_twin
from tests import oracle_binding as orc


def _pack(r, g, b, a):
    return (r.astype(np.uint32) | (g.astype(np.uint32) << 10) | (b.astype(np.uint32) << 20) | (a.astype(np.uint32) << 30))


CUBES = {"3d9": (lambda: cubes.analytic_3d(9), True), "3d33": (lambda: cubes.analytic_3d(33), True),
         "1d3": (lambda: "LUT_1D_SIZE 3\nDOMAIN_MIN 0 0.1 0\nDOMAIN_MAX 1 0.9 2\n0 0 0\n0.25 0.9 0.5\n1 1 0.75\n", False)}


@pytest.mark.parametrize("which", list(CUBES))
def test_rgb10a2_oracle_matches_numpy_twin(which):
    cube_text, is3d = CUBES[which][0](), CUBES[which][1]
    rng = np.random.default_rng(10)
    w, h = 97, 13
    r, g, b = (rng.integers(0, 1024, w * h) for _ in range(3))
    a = rng.integers(0, 4, w * h)
    src = _pack(r, g, b, a).astype("<u4").view(np.uint8).reshape(h, w * 4)
    dst = np.zeros_like(src)
    lut = orc.CubeLut(cube_text)
    assert lut.ok and lut.apply(src, w * 4, dst, w * 4, w, h, "RGB10A2_LE") == 0
    got = dst.view("<u4").reshape(-1)
    vals = np.stack([r, g, b], axis=1)
    scale, offset = lut.domain_scale, lut.domain_offset
    if is3d:
        want = np_twin.colorlut_3d(vals, 1023, lut.rgba(), lut.size, scale, offset)
    else:
        want = np_twin.colorlut_1d(vals, 1023, [lut.table(c) for c in range(3)], lut.size, scale, offset)
    assert np.array_equal(got & 1023, want[:, 0]) and np.array_equal((got >> 10) & 1023, want[:, 1]) and np.array_equal((got >> 20) & 1023, want[:, 2])
    assert np.array_equal(got >> 30, a.astype(np.uint32))


def test_identity_lut_is_identity_on_all_1024_values():
    lut = orc.CubeLut("LUT_1D_SIZE 2\n0 0 0\n1 1 1\n")
    v = np.arange(1024)
    src = _pack(v, v[::-1].copy(), (v * 7) % 1024, v % 4).astype("<u4").view(np.uint8).reshape(1, -1)
    dst = np.zeros_like(src)
    assert lut.apply(src, src.shape[1], dst, src.shape[1], 1024, 1, "RGB10A2_LE") == 0
    assert np.array_equal(dst, src)
