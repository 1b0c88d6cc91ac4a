"""The split replay of a captured multi-stream hipGraph (csrc/graphsplit.cpp, include/miseg_hip.h miseg_graph_split_*; opt-in through
MISEG_GRAPH_SPLIT=1 in runtime/graph.py): same results as the runtime's replay of the same capture."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _graph_mod():
    import __graft_entry__ as ge
    ge.load_package()
    from mi_seg_amd.runtime import graph as G
    return G


def _capture(G, split, body):
    old = G.SPLIT_REPLAY
    G.SPLIT_REPLAY = split
    try:
        g = G._Graph()
        with G._graph_capture(g):
            body()
    finally:
        G.SPLIT_REPLAY = old
    return g


def test_split_replay_of_a_forked_capture_matches_the_runtimes_replay():
    """three chains (two forks from the capture stream, one of them forking again), kernels, device-to-device copies and a memset among the
    nodes, edges in both directions: every piece replayed on its own stream gives what the one multi-stream graph gives, replay after replay"""
    G = _graph_mod()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    n = 1 << 16
    state = {k: torch.zeros(n, device=DEV) for k in "abcdefo"}
    x = torch.zeros(n, device=DEV)

    def body():
        cur = torch.cuda.current_stream()
        a, b, c, d, e, f, o = (state[k] for k in "abcdefo")
        a.copy_(x); a.mul_(2.0).add_(1.0)                      # main
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):                            # side 1: from a
            b.copy_(a); b.mul_(3.0)
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):                        # side 2: from b
                c.copy_(b); c.add_(5.0)
            b.add_(1.0)
        d.copy_(a); d.sub_(4.0)                                # main, beside both
        for _ in range(5):
            d.mul_(1.01)
        cur.wait_stream(s2)
        e.zero_(); e.add_(c).add_(d)                           # main after side 2
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):                            # side 1 again: needs e (main -> side edge in the middle)
            f.copy_(e); f.mul_(b)
        d.add_(1.0)
        cur.wait_stream(s1)
        o.copy_(f); o.add_(d)

    outs = {}
    for split in (False, True):
        g = _capture(G, split, body)
        if split:
            assert g.plan is not None and g.info["lanes"] >= 3 and g.info["side_streams"] >= 1, g.info
            assert g.info["segments"] > g.info["lanes"]       # the lanes were cut at the crossing edges
        res = []
        for it in range(3):
            x.fill_(float(it + 1))
            g.replay()
            torch.cuda.synchronize()
            res.append(state["o"].clone())
        outs[split] = res
        del g
    for r0, r1 in zip(outs[False], outs[True]):
        assert torch.equal(r0, r1)
    x1 = torch.full((n,), 3.0, device=DEV)      # the arithmetic itself, replay 3
    a = x1 * 2 + 1; b = a * 3; c = b + 5; b = b + 1; d = (a - 4) * 1.01 ** 5; e = c + d; f = e * b; o = f + d + 1
    assert torch.allclose(outs[True][2], o, rtol=1e-5)


def test_single_chain_capture_keeps_the_runtimes_replay():
    G = _graph_mod()
    t = torch.zeros(1024, device=DEV)
    g = _capture(G, True, lambda: [t.add_(1.0) for _ in range(10)])
    assert g.plan is None and g.info["lanes"] == 1
    g.replay(); g.replay()
    torch.cuda.synchronize()
    assert float(t[0]) == 20.0      # two replays of ten adds (the capture itself executes nothing)


@pytest.mark.timeout(900)
def test_split_replay_of_the_training_step_matches_the_runtimes_replay():
    """the headline step (C-Swin-UNETR fs=48, 96^3, side branch + deferred weight gradients: 3 lanes, ~11 pieces): logits bit-identical,
    gradient arena to the weight-gradient atomics' order"""
    G = _graph_mod()
    import bench
    from mi_seg_amd.runtime.arena import ParamArena
    model = bench.build_model(torch.bfloat16, "c2")
    params = [p for p in model.parameters() if p.requires_grad]
    x = torch.rand(1, 1, 96, 96, 96, device=DEV)
    cot = torch.randn(1, 6, 96, 96, 96, device=DEV)
    res = {}
    for split in (False, True):
        old = G.SPLIT_REPLAY
        G.SPLIT_REPLAY = split
        try:
            arena = ParamArena(params, torch.bfloat16)
            step = G.GraphedStep(model, x.shape, cot.shape, arena=arena)
            for _ in range(2):
                y = step(x, [0], cot)
            torch.cuda.synchronize()
            if split:
                (g, _), _, _ = step.graphs[next(iter(step.graphs))]
                assert g.plan is not None and g.info["lanes"] >= 2, g.info
            res[split] = (y.detach().clone(), arena.flat.clone())
            del step
            arena.detach()
        finally:
            G.SPLIT_REPLAY = old
    assert torch.equal(res[False][0], res[True][0])
    err = float((res[False][1] - res[True][1]).norm() / res[False][1].norm())
    assert err < 1e-5, err
