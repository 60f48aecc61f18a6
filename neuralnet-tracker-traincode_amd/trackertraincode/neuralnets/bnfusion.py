"""Folding eval-mode BatchNorm into the preceding convolution for export (reference: neuralnets/bnfusion.py:24-63,
used by scripts/export_model.py before the ONNX conversion - SURVEY.md §8 row f4).

Only this graph rewrite is built: the ONNX serialisation itself needs the `onnx` package, which this image does not
have (torch.onnx.export raises "Module onnx is not installed").  The rewrite works on a torch.fx GraphModule of the
plain-torch eval path (`torch_eval_module(net)` wraps a HIP backbone so that fx can trace it)."""
from __future__ import annotations

import copy

import torch
import torch.fx as fx
import torch.nn as nn


def _parent_and_leaf(modules: dict, qualname: str):
    parent, _, leaf = qualname.rpartition(".")
    return modules[parent], leaf


def fuse_convbn(net: fx.GraphModule) -> fx.GraphModule:
    """A copy of `net` in which every BatchNorm2d fed by a Conv2d (with no other consumer) is folded into that conv."""
    net = copy.deepcopy(net)
    modules = dict(net.named_modules())
    for node in list(net.graph.nodes):
        if node.op != "call_module" or type(modules[node.target]) is not nn.BatchNorm2d:
            continue
        src = node.args[0]
        if not isinstance(src, fx.Node) or src.op != "call_module" or type(modules[src.target]) is not nn.Conv2d or len(src.users) > 1:
            continue
        fused = torch.nn.utils.fuse_conv_bn_eval(modules[src.target], modules[node.target])
        parent, leaf = _parent_and_leaf(modules, src.target)
        setattr(parent, leaf, fused)
        modules[src.target] = fused
        node.replace_all_uses_with(src)
        net.graph.erase_node(node)
    net.graph.lint()
    net.delete_all_unused_submodules()
    net.recompile()
    return net


class _TorchEvalPath(nn.Module):
    def __init__(self, backbone):
        super().__init__()
        self.backbone = backbone

    def forward(self, x):
        return self.backbone._forward_torch(x)[0]


def torch_eval_module(backbone: nn.Module) -> nn.Module:
    """The plain-torch eval forward of a HIP backbone (features only) as a module fx.symbolic_trace accepts."""
    if backbone.training:
        raise RuntimeError("call .eval() first: BatchNorm folding uses the running statistics")
    return _TorchEvalPath(backbone)
