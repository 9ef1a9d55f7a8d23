"""CPU: the generation token of the frame session's arrays (df-vo_amd/libs/deep_models/session.py SessionArray) -- what lets
KeypointSampler.kp_selection recognise (copies of) the arrays DeepModel.forward_flow returned, and what makes it NOT
recognise an array somebody wrote to.  The reference copies the arrays it gets (dfvo.py:330-333) and never edits them."""
import importlib

import numpy as np

import __graft_entry__ as g

g.dfvo_amd()
S = importlib.import_module("df-vo_amd.libs.deep_models.session")


def _arr():
    base = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    a = base.view(S.SessionArray)
    a._dfvo_tok = (1, 7, "fwd")
    return a


def test_token_survives_copies_and_full_views():
    a = _arr()
    assert a.copy()._dfvo_tok == (1, 7, "fwd")                 # dfvo.py:330: flows[...].copy()
    assert np.array(a, copy=True, subok=True)._dfvo_tok == (1, 7, "fwd")
    assert a[...]._dfvo_tok == (1, 7, "fwd")
    assert a.view()._dfvo_tok == (1, 7, "fwd")


def test_token_is_dropped_by_anything_that_is_not_the_same_data():
    a = _arr()
    assert getattr(a[0], "_dfvo_tok", None) is None            # another shape: not the buffer
    assert getattr(a.reshape(6, 4), "_dfvo_tok", None) is None
    assert getattr(a.astype(np.float64), "_dfvo_tok", None) is None
    assert getattr(np.asarray(a), "_dfvo_tok", None) is None   # plain ndarray
    assert getattr(a + 1, "_dfvo_tok", None) is None and type(a + 1) is np.ndarray
    b = a.copy()
    b[0, 0, 0] = 5
    assert b._dfvo_tok is None and a._dfvo_tok == (1, 7, "fwd")
    c = a.copy()
    c *= 2
    assert c._dfvo_tok is None and float(c[1, 2, 3]) == 46.0
    d = a.copy()
    np.add(d, 1, out=d)
    assert d._dfvo_tok is None
    e = a.copy()
    e[...] = 0
    assert e._dfvo_tok is None


def test_arithmetic_on_session_arrays_gives_plain_arrays_with_the_right_values():
    a = _arr()
    assert np.array_equal(np.sqrt(a), np.sqrt(np.asarray(a)))
    assert float(a.sum()) == float(np.arange(24).sum())
    assert np.array_equal(a.transpose(1, 2, 0), np.asarray(a).transpose(1, 2, 0))
    assert np.array_equal(np.linalg.norm(a, axis=0), np.linalg.norm(np.asarray(a), axis=0))


def test_randomstate_words_round_trip_and_detect_any_draw():
    """what the session's early pose half is validated with (libs/tracker/_ctx.numpy_rng_words / set_numpy_rng): the 625 words
    are np.random's whole integer stream state -- equal words <=> no draw in between, and handing words back reproduces the
    stream; a pending Gaussian (has_gauss / cached_gaussian) is left alone"""
    ctx = importlib.import_module("df-vo_amd.libs.tracker._ctx")
    np.random.seed(4869)
    np.random.randn()                       # leaves a cached Gaussian behind
    w0 = ctx.numpy_rng_words()
    assert w0.dtype == np.uint32 and w0.shape == (625,) and w0.flags.c_contiguous
    assert np.array_equal(w0, ctx.numpy_rng_words())           # reading the state draws nothing
    st0 = np.random.get_state()
    a = np.random.randint(0, 1 << 30, 5)
    assert not np.array_equal(w0, ctx.numpy_rng_words())       # any draw shows
    np.random.permutation(7)
    ctx.set_numpy_rng(w0)
    st1 = np.random.get_state()
    assert np.array_equal(st1[1], st0[1]) and st1[2] == st0[2] and st1[3] == st0[3] and st1[4] == st0[4]
    assert np.array_equal(np.random.randint(0, 1 << 30, 5), a)
    # a draw that leaves the position unchanged modulo nothing: even one word consumed moves the position
    np.random.seed(1)
    w1 = ctx.numpy_rng_words()
    np.random.random_sample()
    w2 = ctx.numpy_rng_words()
    assert w1[624] != w2[624] or not np.array_equal(w1[:624], w2[:624])
