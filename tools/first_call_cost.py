"""What the FIRST call of a (K, E) shape costs on top of the later ones (q-mer table build, workspaces): wall ms of three calls each."""
import sys, time
sys.path.insert(0, '.')
import torch, genmap_amd as g
from genmap_amd import synth
codes, lens, desc = synth.workload("grch38", 1.0)
t0 = time.time(); ix = g.Index.build(codes, lens, sampling=1); print(f"{desc}: index in {time.time()-t0:.1f} s", flush=True)
out = torch.zeros(len(codes) + 16, dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for K, E, fr in ((30, 0, 1.0), (30, 1, 0.02), (100, 0, 1.0)):
    nk = len(codes) - K + 1
    step = K - g.tuned_infix_length(K, E) + 1
    span = int(nk * fr) // step * step
    rng = None if fr >= 1 else (0, span)
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        ix.map_device(out.data_ptr(), K, E, value_bits=8, kmer_range=rng, stream=st)
        torch.cuda.synchronize(); print(f"K={K} E={E} call {i}: {1e3*(time.time()-t0):.1f} ms wall", flush=True)
print("index info:", ix.info() if hasattr(ix, "info") else "?")
