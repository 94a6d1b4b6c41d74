"""The step entry points of the C ABI only enqueue kernels on the caller's stream: a caller may capture them into a hipGraph
(here through torch.cuda.graph, which hands the capturing stream to the ABI) and replay it.  The context flips its
double-buffered head / fov state on the HOST at every call, so a captured sequence holds an even number of steps
(INTEGRATION.md "Stream capture").  Replayed steps must leave exactly what the same steps launched one by one leave."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(dev, n, seed, pool=4):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    frames = [torch.randint(0, 256, (n, 2, 210, 160, 3), dtype=torch.uint8, generator=g).to(dev) for _ in range(pool)]
    cmds = []
    for _ in range(pool):
        c = torch.full((n,), 2, dtype=torch.uint8)
        c[torch.randint(0, n, (max(1, n // 8),), generator=g)] = 1
        cmds.append(c.to(dev))
    acts = [(torch.rand((n, 2), generator=g) * 65 - 5).to(dev) for _ in range(pool)]
    types = [torch.randint(0, 2, (n,), dtype=torch.int32, generator=g).to(dev) for _ in range(pool)]
    return frames, cmds, acts, types


@pytest.mark.parametrize("kind", ["fixed", "peripheral", "flexible"])
def test_captured_steps_replay_like_eager_steps(kind):
    from active_gym import ObsPipeline
    dev = torch.device("cuda:0")
    n = 33
    kw = dict(num_envs=n, kind=kind, obs_size=(84, 84), frame_stack=4, fov_size=(30, 30), fov_init_loc=(3, 7),
              sensory_action_mode="absolute", resize_to_full=True, device=dev)
    if kind == "peripheral":
        kw["peripheral_res"] = (20, 20)
    frames, cmds, acts, types = _inputs(dev, n, 7)
    if kind == "flexible":
        g = torch.Generator(device="cpu")
        g.manual_seed(11)
        acts = [torch.where(t[:, None].cpu() == 1, torch.randint(10, 61, (n, 2), generator=g).float(), a.cpu()).contiguous().to(dev)
                for a, t in zip(acts, types)]

    def make():
        p = ObsPipeline(**kw)
        obs = torch.empty(p.obs_shape, dtype=torch.float32, device=dev)
        loc = torch.empty((n, 2), dtype=torch.int32, device=dev)
        res = torch.empty((n, 2), dtype=torch.int32, device=dev)

        def step(i):
            p.ingest(frames[i], cmds[i])
            if kind == "flexible":
                p.fovea(acts[i], action_type=types[i], out=obs, loc_out=loc, res_out=res)
            else:
                p.fovea(acts[i], out=obs, loc_out=loc)
        return p, step, obs, loc, res

    pa, sa, oa, la, ra = make()
    pb, sb, ob, lb, rb = make()
    # eager: steps 0 1 | 2 3 0 1 | 2 3 0 1 | 2 3 0 1
    for i in (0, 1):
        sb(i)
    for _ in range(3):
        for i in (2, 3, 0, 1):
            sb(i)
    # captured: two warm steps on the side stream, then a 4-step graph replayed three times
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in (0, 1):
            sa(i)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        for i in (2, 3, 0, 1):
            sa(i)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(pa.stack_u8(), pb.stack_u8())
    assert torch.equal(la, lb)
    if kind == "flexible":
        assert torch.equal(ra, rb)
    assert np.array_equal(oa.cpu().numpy().view(np.uint32), ob.cpu().numpy().view(np.uint32))
    # the context is still usable one launch at a time after the replays, and agrees with the eager one
    sa(2), sb(2)
    torch.cuda.synchronize()
    assert torch.equal(pa.stack_u8(), pb.stack_u8()) and torch.equal(la, lb)
    assert np.array_equal(oa.cpu().numpy().view(np.uint32), ob.cpu().numpy().view(np.uint32))
