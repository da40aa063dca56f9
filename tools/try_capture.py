import torch, faulthandler; faulthandler.enable()
dev = 'cuda:0'
a = torch.ones(1 << 20, device=dev); b = torch.ones(1 << 20, device=dev); c = torch.zeros(1 << 20, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def body(mode):
    main = torch.cuda.current_stream()
    if mode == 'wait_stream':
        s1.wait_stream(main); s2.wait_stream(main)
        with torch.cuda.stream(s1): a.mul_(2.0)
        with torch.cuda.stream(s2): b.mul_(3.0)
        main.wait_stream(s1); main.wait_stream(s2)
        c.copy_(a + b)
    else:
        keep = []
        e = torch.cuda.Event(); keep.append(e); e.record(main); s1.wait_event(e); s2.wait_event(e)
        with torch.cuda.stream(s1): a.mul_(2.0)
        e1 = torch.cuda.Event(); keep.append(e1); e1.record(s1)
        s2.wait_event(e1)                      # cross side-stream dependency
        with torch.cuda.stream(s2): b.mul_(3.0)
        for s in (s1, s2):
            ej = torch.cuda.Event(); keep.append(ej); ej.record(s); main.wait_event(ej)
        c.copy_(a + b)
        return keep
for mode in ('wait_stream', 'events'):
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        k = body(mode)
    g.replay(); torch.cuda.synchronize()
    print(mode, 'ok', float(c[0]))
