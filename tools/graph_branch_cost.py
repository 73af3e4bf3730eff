#!/usr/bin/env python3
"""What a parallel branch costs inside a replayed hipGraph (one MI355X): a chain of short dependent kernels with and without side
branches that are forked off and joined back through events (stream capture, as a captured training step would do it).
    python tools/graph_branch_cost.py
Prints us per replay for: the plain chain | the chain + K side kernels appended to it serially | the chain with the same K kernels on
ONE side branch (one fork, one join) | with each of them on a branch of its own (K forks, K joins)."""
import time
import torch

dev = torch.device("cuda:0")
N_MAIN, K_SIDE = 40, 8
main_bufs = [torch.zeros(1 << 20, device=dev) for _ in range(2)]        # 4 MB each: ~3-4 us per add kernel
side_bufs = [torch.zeros(1 << 18, device=dev) for _ in range(K_SIDE)]


def main_kernel(i):
    main_bufs[(i + 1) & 1].add_(main_bufs[i & 1], alpha=1.0)             # dependent chain


def capture(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(None)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn(torch.cuda.current_stream())
    return g


def plain(st):
    for i in range(N_MAIN):
        main_kernel(i)


def serial(st):
    for i in range(N_MAIN):
        main_kernel(i)
        if 10 <= i < 10 + K_SIDE:
            side_bufs[i - 10].add_(1.0)


def one_branch(st):
    side = torch.cuda.Stream() if st is not None else None
    for i in range(N_MAIN):
        main_kernel(i)
        if i == 10 and side is not None:
            side.wait_stream(st)
            with torch.cuda.stream(side):
                for b in side_bufs:
                    b.add_(1.0)
        elif i == 10:
            for b in side_bufs:
                b.add_(1.0)
    if side is not None:
        st.wait_stream(side)


def many_branches(st):
    sides = []
    for i in range(N_MAIN):
        main_kernel(i)
        if 10 <= i < 10 + K_SIDE:
            if st is None:
                side_bufs[i - 10].add_(1.0)
            else:
                s = torch.cuda.Stream()
                s.wait_stream(st)
                with torch.cuda.stream(s):
                    side_bufs[i - 10].add_(1.0)
                sides.append(s)
    for s in sides:
        st.wait_stream(s)


def timeit(g, n=300):
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for name, fn in (("plain chain of %d" % N_MAIN, plain), ("+ %d kernels serially" % K_SIDE, serial), ("+ %d on ONE side branch" % K_SIDE, one_branch),
                 ("+ %d on %d branches" % (K_SIDE, K_SIDE), many_branches)):
    g = capture(fn)
    print(f"{name:28s} {timeit(g):8.1f} us per replay")
