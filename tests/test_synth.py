import hashlib

from xpng_amd.synth import synth_raster, to_seven_bytes


def test_small_synthetic_md5s_are_stable(manifest):
    for name in ("synth_photo_700x500_rgba", "synth_noise_64x64_rgb", "synth_gray_445x444_rgb", "synth_flat_5x7_rgba"):
        e = manifest[name]
        kind = e["src"][6:]
        r = synth_raster(kind, e["w"], e["h"], e["ch"] == 4)
        assert hashlib.md5(to_seven_bytes(r)).hexdigest() == e["seven_md5"]


def test_band_generation_matches_whole():
    import numpy as np
    whole = synth_raster("photo", 300, 200, True)
    band = synth_raster("photo", 300, 50, True, y0=100)
    assert np.array_equal(whole[100:150], band)
    col = synth_raster("noise", 40, 200, False, x0=17)
    assert np.array_equal(synth_raster("noise", 300, 200, False)[:, 17:57], col)
