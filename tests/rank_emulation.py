"""ONE process plays every rank of a data-parallel loss in turn (test helper; used on the GPU at BASELINE config 3 / 5 sizes by
tests/test_parity_at_size_gpu.py and validated on CPU against plain autograd by tests/test_loss_rank_emulation.py)."""
import torch


class Collectives:
    """stands in for the process group inside open_clip_amd.loss while ONE process plays every rank in turn: the all-gather hands out the
    gathered features, reduce-scatter / all-reduce RECORD what this rank contributes and leave the result at zero / untouched -- the test then
    forms the sums over the ranks' recorded contributions itself (both collectives are plain sums, and the loss adds their result linearly)."""

    def __init__(self, gathered):
        self.gathered, self.rs_inputs, self.ar_inputs = gathered, [], []

    def all_gather(self, out, inp, comm=None):
        out.copy_(self.gathered.reshape(out.shape))

    def reduce_scatter(self, out, inp, comm=None):
        self.rs_inputs.append(inp.detach().clone())
        out.zero_()

    def all_reduce(self, t, comm=None):
        self.ar_inputs.append(t.detach().clone())


def unit_features(N, E, seed, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    I = torch.nn.functional.normalize(torch.randn(N, E, device=dev, generator=g), dim=-1)
    # texts correlated with their own image (a trained model's diagonal stands out; at cos ~ N(0, 1/E) alone the softmax is nearly flat)
    T = torch.nn.functional.normalize(0.6 * I + 0.8 * torch.nn.functional.normalize(torch.randn(N, E, device=dev, generator=g), dim=-1), dim=-1)
    return I.contiguous(), T.contiguous()
