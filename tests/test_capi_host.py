"""C-ABI library: loads, exports every declared symbol, host-side table preparation matches the oracle (no GPU)."""
import os
import re

import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import engine_kwargs, make_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "acmpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(acmpc_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from acmpc_amd import _capi
    lib = _capi.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), "libacmpc_hip.so does not export %s" % name
    assert set(declared) == set(_capi.SIGNATURES), "ctypes binding and header disagree"
    assert lib.acmpc_version().startswith(b"acmpc-hip")


def test_params_struct_matches_header_layout():
    import ctypes
    from acmpc_amd import _capi
    # 10 x 4-byte ints then 18 doubles, no padding surprises
    assert ctypes.sizeof(_capi.Params) == 40 + 18 * 8
    assert _capi.record_floats(49) == _capi.load_library().acmpc_record_floats(49) == 4 + 98 + 150


@pytest.mark.parametrize("track,H", [("monza", 20), ("monza", 50), ("nordschleife", 80), ("spa", 50)])
def test_host_tables_match_oracle(track, H):
    """acmpc_set_paths restates linearise() + the corridor bounds (dynamics.py:65-103, control.py:57-60) in C++
    double precision: the packed float32 rows must equal the oracle's bit for bit (mode S) and to one ulp in the
    cos/sin columns of mode T (libm vs NumPy)."""
    from acmpc_amd import Engine
    probs = [make_problem(orc, track, H, 4, seed=s) for s in range(3)]
    n = H - 1
    tables = np.stack([p["table"] for p in probs])
    eng = Engine(**engine_kwargs(probs[0], 0, 3, 4, n))
    eng.set_paths(tables)
    for p, prob in enumerate(probs):
        np.testing.assert_array_equal(eng.coefficients(p), orc.coefficients_spatial(prob["table"], prob["limits"].margin))
    eng = Engine(**engine_kwargs(probs[0], 1, 3, 4, n))
    eng.set_paths(tables)
    for p, prob in enumerate(probs):
        got, want = eng.coefficients(p), orc.coefficients_temporal(prob["table"], prob["limits"].margin)
        exact = [orc.CT_X, orc.CT_Y, orc.CT_PSI, orc.CT_KREF, orc.CT_VREF, orc.CT_HALF]
        np.testing.assert_array_equal(got[:, exact], want[:, exact])
        np.testing.assert_allclose(got[:, [orc.CT_COS, orc.CT_SIN]], want[:, [orc.CT_COS, orc.CT_SIN]], rtol=0,
                                   atol=6e-8)


def test_key_packing_orders_like_cost_then_index():
    from acmpc_amd import _capi
    rng = np.random.default_rng(0)
    costs = np.concatenate([rng.normal(0, 10, 200), [0.0, -0.0, 1e-30, -1e-30, np.inf, 3.0, 3.0]]).astype(np.float32)
    idx = rng.integers(0, 2**32 - 1, len(costs), dtype=np.uint64)
    keys = [_capi.pack_key(c, int(i)) for c, i in zip(costs, idx)]
    order = sorted(range(len(costs)), key=lambda j: keys[j])
    want = sorted(range(len(costs)), key=lambda j: (float(costs[j]) + 0.0 if costs[j] != 0 else
                                                    (-0.0 if np.signbit(costs[j]) else 0.0), int(idx[j])))
    # -0.0 orders just below +0.0 in the key; otherwise (cost, index) lexicographic
    assert [(float(costs[j]), int(idx[j])) for j in order if costs[j] != 0] == \
           [(float(costs[j]), int(idx[j])) for j in want if costs[j] != 0]
    for c, i, k in zip(costs, idx, keys):
        assert _capi.key_index(k) == int(i)
        assert _capi.key_cost(k) == c or (np.isnan(c) and np.isinf(_capi.key_cost(k)))
    assert _capi.key_cost(_capi.pack_key(float("nan"), 7)) == np.inf
    assert _capi.key_cost(_capi.pack_key(float("-inf"), 7)) == np.inf  # every non-finite cost ranks last
    assert _capi.pack_key(1.0, 5) < _capi.pack_key(1.0, 6) < _capi.pack_key(np.nextafter(np.float32(1), 2), 0)
    assert _capi.pack_key(-2.0, 9) < _capi.pack_key(-1.0, 0) < _capi.pack_key(0.0, 0)


def test_errors_without_device_or_tables():
    """Host-only error paths; on a box without a GPU the compute call must fail loudly (ENODEVICE), never fall
    back to a CPU path."""
    import torch
    from acmpc_amd import Engine, EngineError
    prob = make_problem(orc, "monza", 20, 8, seed=0)
    with pytest.raises(EngineError):
        Engine(**engine_kwargs(prob, 7, 1, 8, 19))  # unknown mode
    with pytest.raises(EngineError):
        Engine(**engine_kwargs(prob, 0, 1, 8, 5000))  # max_steps beyond the LDS staging limit
    with pytest.raises(EngineError):
        Engine(**engine_kwargs(prob, 0, 70000, 8, 19))  # more problems than the grid's y dimension holds
    eng = Engine(**engine_kwargs(prob, 0, 1, 8, 19))
    with pytest.raises(EngineError) as e:
        eng.coefficients(0)
    assert e.value.code == -5
    with pytest.raises(EngineError) as e:
        eng.set_paths(np.zeros((2, 7, 19)))
    assert e.value.code == -4
    eng.set_paths(prob["table"])
    with pytest.raises(EngineError) as e:
        eng.reduce_across_ranks(0, 0, 1)         # no communicator, no keys: rejected before RCCL is looked up
    assert e.value.code == -1
    if not torch.cuda.is_available():
        with pytest.raises(EngineError) as e:
            eng.solve(prob["x0"][None], prob["U"][None])
        assert e.value.code == -3 and "no CPU fallback" in str(e.value)


def test_philox_known_answers_and_oracle_agreement():
    """Random123's published known-answer vectors for philox4x32-10, and library == NumPy restatement."""
    from acmpc_amd import _capi
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        np.testing.assert_array_equal(_capi.philox4x32(ctr, key), np.array(want, dtype=np.uint32))
        np.testing.assert_array_equal(orc.philox4x32_10(np.array(ctr), np.array(key)), np.array(want, dtype=np.uint32))
    rng = np.random.default_rng(3)
    ctr = rng.integers(0, 2**32, (50, 4), dtype=np.uint64)
    key = rng.integers(0, 2**32, (50, 2), dtype=np.uint64)
    got = orc.philox4x32_10(ctr, key)
    for i in range(50):
        np.testing.assert_array_equal(_capi.philox4x32(ctr[i], key[i]), got[i])


def test_key_minimum_is_the_oracles_argmin_on_arbitrary_bit_patterns():
    """Property (hypothesis): for ANY float32 bit patterns - NaNs of either sign, infinities, denormals, signed
    zeros - the index inside the smallest packed key is the oracle's pick_best (np.argmin's first minimum with
    non-finite costs ranked last) and the C restatement's argmin."""
    import c_oracle
    from hypothesis import given, settings, strategies as st
    from acmpc_amd import _capi

    @settings(max_examples=300, deadline=None, derandomize=True)
    @given(st.lists(st.integers(min_value=0, max_value=2**32 - 1), min_size=1, max_size=40))
    def check(bits):
        costs = np.array(bits, dtype=np.uint32).view(np.float32)
        keys = [_capi.pack_key(float(c), j) for j, c in enumerate(costs)]
        best = _capi.key_index(min(keys))
        # signed zeros: the key ranks -0.0 just below +0.0, np.argmin treats them as equal -> compare on values
        want = orc.pick_best(costs)[0]
        assert best == want or (costs[best] == 0 and costs[want] == 0)
        assert c_oracle.argmin(costs) == want or (costs[best] == 0 and costs[want] == 0)

    check()



def test_unpack_decision_temporal_on_the_host():
    """acmpc_unpack_decision_temporal (no GPU): a mode T plan's dec.x -> poses as the prediction, time = i dt, the
    derivatives of the plan's own controls."""
    from acmpc_amd import _capi
    n, dt, wheelbase = 19, 0.05, 2.65
    rng = np.random.default_rng(3)
    poses = rng.normal(size=(n + 1, 3))
    controls = np.stack([rng.uniform(8, 28, n), rng.uniform(-0.05, 0.05, n)], axis=1)
    z = np.concatenate([poses.ravel(), controls.ravel()])
    projected, prediction, cum_time, times, accelerations, steer_rates = _capi.unpack_decision_temporal(z, n, dt, wheelbase)
    np.testing.assert_array_equal(projected[0], controls[:, 0])
    np.testing.assert_allclose(projected[1], np.arctan(controls[:, 1] * wheelbase), rtol=4e-16, atol=0)   # libm vs NumPy atan
    np.testing.assert_array_equal(prediction, poses[:n, :2])
    np.testing.assert_array_equal(cum_time, np.arange(n) * dt)
    np.testing.assert_array_equal(times, np.full(n - 1, dt))
    np.testing.assert_array_equal(accelerations, (projected[0, 1:] - projected[0, :-1]) / dt)
    np.testing.assert_array_equal(steer_rates, (projected[1, 1:] - projected[1, :-1]) / dt)
    with pytest.raises(ValueError):
        _capi.unpack_decision_temporal(z[:-1], n, dt, wheelbase)


def test_rccl_communicator_helpers_refuse_bad_arguments():
    """acmpc_rccl_unique_id / acmpc_rccl_comm_create / acmpc_rccl_comm_destroy (the communicator bench.py --collective capi
    makes for acmpc_reduce_across_ranks): the refusals that need neither a GPU nor RCCL."""
    import ctypes as C
    from acmpc_amd import _capi
    lib = _capi.load_library()
    assert lib.acmpc_rccl_unique_id(None) == _capi.EINVAL
    comm = C.c_void_p(123)
    identifier = C.create_string_buffer(_capi.RCCL_UNIQUE_ID_BYTES)
    for n_ranks, rank in ((0, 0), (2, 2), (2, -1)):
        assert lib.acmpc_rccl_comm_create(identifier, n_ranks, rank, -1, C.byref(comm)) == _capi.EINVAL
    assert lib.acmpc_rccl_comm_create(None, 1, 0, -1, C.byref(comm)) == _capi.EINVAL
    assert lib.acmpc_rccl_comm_create(identifier, 1, 0, -1, None) == _capi.EINVAL
    assert lib.acmpc_rccl_comm_destroy(None) == _capi.OK          # nothing to destroy
    with pytest.raises(_capi.EngineError) as refused:
        _capi.rccl_comm_create(b"too short", 1, 0)
    assert refused.value.code == _capi.EINVAL
