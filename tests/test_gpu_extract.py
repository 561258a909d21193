"""GPU parity, seam 1: SpinningSensorKeypointExtractor::ComputeKeyPoints through the C ABI against the
CPU oracle and the golden vectors.  Bar: bit-exact -- the four score arrays (float bit patterns), the
validity / label bitsets and the keypoint clouds (index sets AND order)."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def assert_extraction_equal(ctx, ex, O, L, pts, params=None):
    ctx.upload_frame(pts)
    counts = ctx.extract_keypoints(params)
    ref = ex.compute(pts, params)
    assert counts.tolist() == ref.tolist()
    for i, name in enumerate(O.DEBUG_NAMES):
        a, b = ctx.debug_array(i), ex.debug(i)
        bad = np.flatnonzero(bits(a) != bits(b))
        assert bad.size == 0, f"{name}: {bad.size} mismatches, first at {bad[:5]}: gpu {a[bad[:5]]} oracle {b[bad[:5]]}"
    for k in range(3):
        assert ctx.keypoints(L.SET_RAW_CURRENT, k).tobytes() == ex.keypoints(k).tobytes(), f"keypoint cloud {k}"
    return counts


@pytest.mark.parametrize("model,frames", [(8, 3), (16, 2), (64, 1), (128, 2)])
def test_extraction_bit_exact(gpu_ctx, O, L, model, frames):
    """BASELINE.json configs: VLP-16, HDL-64 and the full-size VLS-128 scan (~260k points)."""
    ex = O.Extractor()
    gpu_ctx.azimuthal_resolution = 0.0
    for f in range(frames):
        pts, _ = L.synth_frame(model, 1000 + model, f)
        c = assert_extraction_equal(gpu_ctx, ex, O, L, pts)
        assert c[1] > 0
    assert np.float32(gpu_ctx.azimuthal_resolution) == np.float32(ex.azimuthal_resolution)


def test_extraction_matches_golden_vectors(gpu_ctx, L, golden):
    gpu_ctx.azimuthal_resolution = 0.0
    for f in range(4):
        gpu_ctx.upload_frame(golden[f"frame{f}"])
        counts = gpu_ctx.extract_keypoints()
        for i in range(10):
            assert np.array_equal(bits(gpu_ctx.debug_array(i)), bits(golden[f"dbg{f}"][i])), (f, i)
        for k in range(3):
            assert counts[k] == golden[f"kp{f}_{k}"].size
            assert gpu_ctx.keypoints(L.SET_RAW_CURRENT, k).tobytes() == golden[f"kp{f}_{k}"].tobytes()


@pytest.mark.parametrize(
    "kw",
    [
        dict(neighbor_width=3),
        dict(neighbor_width=5),
        dict(neighbor_width=6, edge_sin_angle_threshold=0.7),
        dict(min_distance_to_sensor=4.0, min_beam_surface_angle=20.0),
        dict(plane_sin_angle_threshold=0.2, edge_depth_gap_threshold=0.5, edge_saliency_threshold=0.5, edge_intensity_gap_threshold=5.0),
        dict(dist_to_line_threshold=0.05),
    ],
)
def test_extraction_non_default_parameters(gpu_ctx, O, L, kw):
    pts, _ = L.synth_frame(16, 1003, 1)
    ex = O.Extractor()
    ex.azimuthal_resolution = gpu_ctx.azimuthal_resolution = 0.0035
    assert_extraction_equal(gpu_ctx, ex, O, L, pts, L.ExtractParams(**kw))


def test_extraction_ragged_and_degenerate_inputs(gpu_ctx, O, L):
    pts, _ = L.synth_frame(16, 1000, 0)
    ex = O.Extractor()
    ex.azimuthal_resolution = gpu_ctx.azimuthal_resolution = 0.0035
    rng = np.random.default_rng(0)
    cases = {
        "single point": pts[:1],
        "five points": pts[:5],
        "ring with fewer than 2W+1 points": np.concatenate([pts[pts["laser_id"] == 2][:8], pts[pts["laser_id"] == 5]]),
        "only the top ring (rings below are empty)": pts[pts["laser_id"] == 15],
        "random 40% dropout (ragged rings)": pts[rng.random(pts.size) > 0.4],
        "rings delivered one after the other instead of interleaved": pts[np.argsort(pts["laser_id"], kind="stable")],
        "dual returns: every point twice": np.repeat(pts[pts["laser_id"] < 4], 2),
        "identical points (zero covariance)": np.repeat(pts[100:101], 300),
    }
    for name, c in cases.items():
        assert_extraction_equal(gpu_ctx, ex, O, L, np.ascontiguousarray(c))
    # sparse laser ids with gaps, up to the supported maximum
    sparse = pts.copy()
    sparse["laser_id"] = sparse["laser_id"] * 30 + 7
    assert_extraction_equal(gpu_ctx, ex, O, L, sparse)
    assert gpu_ctx.nb_laser_rings() >= 458


def test_extraction_rejects_unsupported_input_loudly(gpu_ctx, L):
    pts, _ = L.synth_frame(8, 1000, 0)
    bad = pts.copy()
    bad["laser_id"][10] = 600  # >= 512 rings is outside the supported range: error, not silence
    gpu_ctx.upload_frame(bad)
    with pytest.raises(L.LsaError):
        gpu_ctx.extract_keypoints()
    with pytest.raises(L.LsaError):
        gpu_ctx.upload_frame(pts)
        gpu_ctx.extract_keypoints(L.ExtractParams(neighbor_width=9))
    gpu_ctx.upload_frame(pts)
    gpu_ctx.extract_keypoints()  # the context stays usable


def test_previous_keypoints_follow_the_reference_swap(gpu_ctx, L):
    a, _ = L.synth_frame(8, 1000, 0)
    b, _ = L.synth_frame(8, 1000, 1)
    gpu_ctx.upload_frame(a)
    gpu_ctx.extract_keypoints()
    first = [gpu_ctx.keypoints(L.SET_RAW_CURRENT, k) for k in range(3)]
    gpu_ctx.upload_frame(b)
    gpu_ctx.extract_keypoints()
    for k in range(3):  # PreviousRawKeypoints = CurrentRawKeypoints (Slam.cxx:751)
        assert gpu_ctx.keypoints(L.SET_RAW_PREVIOUS, k).tobytes() == first[k].tobytes()


def test_frame_store_gives_the_same_result_as_upload(gpu_ctx, L):
    pts, _ = L.synth_frame(16, 1000, 3)
    gpu_ctx.upload_frame(pts)
    c1 = gpu_ctx.extract_keypoints()
    k1 = [gpu_ctx.keypoints(L.SET_RAW_CURRENT, k) for k in range(3)]
    gpu_ctx.store_frame(3, pts)
    gpu_ctx.use_stored_frame(3)
    c2 = gpu_ctx.extract_keypoints()
    assert c1.tolist() == c2.tolist()
    for k in range(3):
        assert gpu_ctx.keypoints(L.SET_RAW_CURRENT, k).tobytes() == k1[k].tobytes()


def test_device_arithmetic_is_bit_identical_to_the_host(gpu_ctx, O):
    """H1 of SURVEY.md: portable trig, IEEE sqrt and division must agree bit for bit."""
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-4, 4, 100000), rng.uniform(0, np.pi / 3, 100000), 10.0 ** rng.uniform(-30, 30, 50000)])
    y = np.concatenate([rng.uniform(-4, 4, 100000), rng.uniform(0, 1, 100000), 10.0 ** rng.uniform(-30, 30, 50000)])
    for fn in range(9):
        xs, ys = (np.abs(x) if fn in (3, 5) else x), y
        if fn in (7, 8):  # lsa_asin / lsa_acos: [-1, 1] with both ends, the neighbourhood of +-1 / 0.5 / 0 densely
            xs = np.concatenate([rng.uniform(-1, 1, 200000), 1 - 10.0 ** rng.uniform(-17, 0, 25000), -1 + 10.0 ** rng.uniform(-17, 0, 24996), [1., -1., 0., 0.5]])
        if fn in (0, 1):  # lsa_sin / lsa_cos are specified for |x| < 1e5 (angles here are <= pi)
            xs = np.where(np.abs(xs) < 1e4, xs, np.fmod(xs, 1e4))
        if fn in (3, 4):  # float: stay inside the normal range (no denormal results)
            xs, ys = np.clip(np.abs(xs), 1e-15, 1e15) * np.sign(xs + 1e-300), np.clip(np.abs(ys), 1e-15, 1e15)
        a, b = gpu_ctx.selftest_math(fn, xs, ys), O.math(fn, xs, ys)
        assert np.array_equal(bits(a), bits(b)), f"fn {fn}: {np.count_nonzero(bits(a) != bits(b))} results differ"


def test_unused_keypoint_types_come_out_empty(O, L):
    """Slam::ExtractKeypoints drops the types the caller does not use (Slam.cxx:789-793); the time range that rides
    on the extraction (Slam::InitUndistortion, Slam.cxx:1291-1300) then only covers the types that are kept."""
    pts, _ = L.synth_frame(16, 1000, 2)
    ex = O.Extractor()
    ex.compute(pts)
    want = [ex.keypoints(k) for k in range(3)]
    lib = L.lib()
    gpu_ctx = L.Context(0)  # a context of its own: the azimuthal resolution is estimated on the first frame it sees
    try:
        for mask in (3, 5, 2, 0, 7):
            assert lib.lsa_set_keypoint_types(gpu_ctx.h, mask) == 0
            gpu_ctx.upload_frame(pts)
            counts = gpu_ctx.extract_keypoints()
            kept = [want[k] if (mask >> k) & 1 else want[k][:0] for k in range(3)]
            assert counts.tolist() == [a.size for a in kept]
            for k in range(3):
                assert gpu_ctx.keypoints(L.SET_RAW_CURRENT, k).tobytes() == kept[k].tobytes()
            gpu_ctx.reset_working_keypoints()
            t0, t1 = gpu_ctx.working_time_range()
            times = np.concatenate([a["time"] for a in kept])
            if times.size:
                assert (t0, t1) == (times.min(), times.max())
            else:
                assert t0 > t1  # the reference's untouched initial values
    finally:
        gpu_ctx.close()
