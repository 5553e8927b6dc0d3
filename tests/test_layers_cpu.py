"""Host-side behaviour of gcn_amd.layers that needs no GPU."""
import torch

import gcn_amd


def test_reference_format_checkpoint_loads_strictly():
    """a checkpoint of the reference's model holds gc1.weight, gc1.bias, gc2.weight, gc2.bias and nothing else
    (pygcn/gcn6.py:66-97,201-236; profiling_gcn.py:165-170 saves and loads exactly that): strict loading must work,
    and the fused-epilogue dropout state starts fresh"""
    model = gcn_amd.GCN(12, 8, 3, dataset="synthetic", device="cpu", order=None)
    ref = {"gc1.weight": torch.randn(12, 8), "gc1.bias": torch.randn(8), "gc2.weight": torch.randn(8, 3), "gc2.bias": torch.randn(3)}
    missing, unexpected = model.load_state_dict(dict(ref), strict=True)
    assert not missing and not unexpected
    assert torch.equal(model.gc1.weight, ref["gc1.weight"]) and torch.equal(model.gc2.bias, ref["gc2.bias"])
    assert model.dropout_seed is None and model._dropout_calls == 0
    # the model's own state still round-trips with its extra state
    model.dropout_seed, model._dropout_calls = 1234, 7
    sd = model.state_dict()
    assert sd["_extra_state"] == {"dropout_seed": 1234, "dropout_calls": 7}
    other = gcn_amd.GCN(12, 8, 3, dataset="synthetic", device="cpu", order=None)
    other.load_state_dict(sd, strict=True)
    assert other.dropout_seed == 1234 and other._dropout_calls == 7
    # as a sub-module (prefix) too
    class Wrap(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = gcn_amd.GCN(12, 8, 3, dataset="synthetic", device="cpu", order=None)
    w = Wrap()
    w.load_state_dict({"net." + k: v for k, v in ref.items()}, strict=True)
    assert torch.equal(w.net.gc2.weight, ref["gc2.weight"])
