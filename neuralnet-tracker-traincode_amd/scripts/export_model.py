#!/usr/bin/env python
"""Export surface of the pose estimator (reference: scripts/export_model.py) - the step AFTER the training path.

What the reference's script does, and what is built here:

  clear_denormals          :36-50    weights below 1e-20 in magnitude -> 0 (denormals are slow on CPUs)          built, same threshold
  ModelForOpenTrack        :116-146  output selection / order / names OpenTrack's neuralnet tracker binds:        built, same names
                                     x -> (pos_size, quat, box[, pos_size_scales_tril, rotaxis_scales_tril])
  ExportModel              :149-169  every output of the network, in the network's own key order                  built
  convert_posemodel_onnx   :201-279  denormal flush -> wrapper -> eval -> torch.onnx.export(opset 13, constant
                                     folding, dynamic batch axis for the complete model) -> onnx shape inference,
                                     onnxsim, checker, model_version 4, optional fp16 -> onnxruntime comparison     see below
  quantize_backbone        :53-113   torch.ao post-training quantisation over 20 training batches                  NOT built

This image has no `onnx` / `onnxsim` / `onnxruntime`, so `torch.onnx.export` cannot serialise.  `convert_posemodel_onnx` therefore
runs everything up to the serialisation - flush, wrapper, the plain-torch CPU eval path of the HIP network (optionally with every
BatchNorm folded into its convolution, neuralnets/bnfusion.py), `torch.jit.trace` of exactly what would be exported, a numerical
check of the traced (and folded) module against the eager one - and writes `<name>[_complete].pt` (TorchScript) plus
`<name>[_complete].contract.json` (input / output names, shapes, dynamic axes, opset, model_version, doc_string: what the ONNX file
would declare).  When the `onnx` package is importable the same function goes on to `torch.onnx.export` with the reference's
arguments.  The output contract is pinned to the reference's own wrappers by tests/golden/export_contract.npz.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from os.path import dirname, splitext

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, dirname(dirname(os.path.abspath(__file__))))

import trackertraincode.neuralnets.models  # noqa: E402
from trackertraincode.neuralnets.bnfusion import fuse_convbn, torch_eval_module  # noqa: E402

OPSET_VERSION = 13      # reference :243
MODEL_VERSION = 4       # reference :257
DOC_STRING = "Head pose prediction"  # reference :256


def clear_denormals(state_dict, threshold=1.0e-20, verbose=True):
    """Copy of `state_dict` whose float32 entries below `threshold` in magnitude are zero (reference :36-50: real denormals start
    below 2e-38; the threshold was tuned on CPU inference time)."""
    state_dict = {k: v.detach().clone() for k, v in state_dict.items()}
    if verbose:
        print("Denormals or zeros:")
    for k, v in state_dict.items():
        if v.dtype == torch.float32:
            mask = torch.abs(v) > threshold
            n = int(torch.count_nonzero(~mask))
            if n and verbose:
                print(f"{k:40s}: {n:10d} ({n / max(v.numel(), 1) * 100}%)")
            v *= mask.to(torch.float32)
    return state_dict


class ModelForOpenTrack(nn.Module):
    """Rearranges the model output into what OpenTrack binds (reference :116-146)."""

    def __init__(self, original):
        super().__init__()
        self._original = original
        self.input_names = ["x"]
        self._output_name_map = [("coord", "pos_size"), ("pose", "quat"), ("roi", "box")]
        if original.enable_uncertainty:
            self._output_name_map += [("coord_scales", "pos_size_scales_tril"), ("pose_scales_tril", "rotaxis_scales_tril")]

    @property
    def output_names(self):
        return [n for _, n in self._output_name_map]

    @property
    def input_resolution(self):
        return self._original.input_resolution

    def forward(self, x):
        y = self._original(x)
        return tuple(y[k] for k, _ in self._output_name_map)


class ExportModel(nn.Module):
    """Every output of the network as a tuple, in the network's own key order (reference :149-169)."""

    def __init__(self, original: nn.Module):
        super().__init__()
        self._original = original
        self.input_names = ["x"]
        self.output_names = ExportModel._compute_output_names(original)

    @staticmethod
    def _compute_output_names(original):
        original.eval()
        with torch.no_grad():
            y = original(torch.zeros((1, 1, original.input_resolution, original.input_resolution)))
        return list(y.keys())

    @property
    def input_resolution(self):
        return self._original.input_resolution

    def forward(self, x):
        y = self._original(x)
        return tuple(_as_tensor(y[k]) for k in self.output_names)


def _as_tensor(v):
    return v if isinstance(v, torch.Tensor) else v.value  # rotation containers (QuatRepr / Mat33Repr) export their tensor


def fold_batchnorm(net):
    """`net` (a CPU eval-mode NetworkWithPointHead) with every BatchNorm of its backbone folded into the preceding convolution
    (reference: neuralnets/bnfusion.py:24-63, applied before quantisation :101-103; a plain ONNX export gets the same folding from the
    exporter's constant folding).  Returns a module with the same forward signature whose backbone is the folded fx graph."""
    import copy

    import torch.fx as fx

    net = copy.deepcopy(net).eval()
    folded = fuse_convbn(fx.symbolic_trace(torch_eval_module(net.convnet)))

    class _FoldedBackbone(nn.Module):
        def __init__(self, graph, like):
            super().__init__()
            self.graph = graph
            self.num_features = like.num_features

        def forward_features(self, x):
            return self.graph(x)

        def forward(self, x):
            return self.graph(x), None

    net.convnet = _FoldedBackbone(folded, net.convnet)
    return net


def destination_of(filename, for_opentrack, quantize=False, fp16=False, ext=".onnx"):
    """File name rule of the reference (:221-229)."""
    destination = splitext(filename)[0]
    if quantize:
        destination += "_ptq"
    if fp16:
        destination += "_fp16"
    if not for_opentrack:
        destination += "_complete"
    return destination + ext


@torch.no_grad()
def convert_posemodel_onnx(net: nn.Module, filename, for_opentrack=True, quantize=False, fp16=False, fold=True, check_tol=1.0e-5):
    """Reference :201-279.  Returns the contract dict that was written next to the exported file."""
    if quantize:
        raise NotImplementedError("post-training quantisation (torch.ao over 20 training batches, reference :53-113) is not built: it needs the "
                                  "HDF5 training sets; see SURVEY.md §8 f4")
    if fp16:
        # (round-3 advisor finding: the flag used to change only the file name and the contract - an fp32 graph labelled fp16)
        raise NotImplementedError("fp16 conversion (onnxconverter_common.float16 + onnxsim + an onnxruntime comparison, reference :243-279) is not "
                                  "built: the image has none of the three packages; export fp32 and convert where they are installed")
    net = net.to("cpu")
    net.load_state_dict(clear_denormals(net.state_dict()))
    wrapped = ModelForOpenTrack(net) if for_opentrack else ExportModel(net)
    wrapped.eval()
    # Batch size: OpenTrack needs 1; otherwise something larger to see that it works more generally (reference :213-216)
    B = 1 if for_opentrack else 5
    inputs = (torch.randn(B, 1, wrapped.input_resolution, wrapped.input_resolution),)
    eager = wrapped(*inputs)
    # what the exporter would serialise: the traced graph of the wrapper (BatchNorm folded like the exporter's constant folding does)
    target = wrapped
    if fold:
        folded_net = fold_batchnorm(net)
        target = (ModelForOpenTrack(folded_net) if for_opentrack else _ExportLike(folded_net, wrapped.output_names)).eval()
    traced = torch.jit.trace(target, inputs, check_trace=False)
    got = traced(*inputs)
    worst = 0.0
    for a, b, name in zip(eager, got, wrapped.output_names):
        a, b = _as_tensor(a), _as_tensor(b)
        delta = float((a - b).abs().max()) / max(float(a.abs().max()), 1.0)
        worst = max(worst, delta)
        if delta > check_tol:
            raise RuntimeError(f"exported graph differs from the eager network in output {name} by {delta:.2e}")
    dynamic_axes = None if for_opentrack else {k: {0: "batch"} for k in (wrapped.input_names + wrapped.output_names)}
    contract = {
        "doc_string": DOC_STRING, "model_version": MODEL_VERSION, "opset_version": OPSET_VERSION,
        "inputs": [{"name": "x", "shape": list(inputs[0].shape), "dtype": "float32"}],
        "outputs": [{"name": n, "shape": list(_as_tensor(v).shape), "dtype": "float32"} for n, v in zip(wrapped.output_names, eager)],
        "dynamic_axes": dynamic_axes, "batchnorm_folded": bool(fold), "max_rel_delta_traced_vs_eager": worst,
        "for_opentrack": bool(for_opentrack), "fp16": bool(fp16),
    }
    try:
        import onnx  # noqa: F401
        have_onnx = True
    except ImportError:
        have_onnx = False
    if have_onnx:  # the reference's call (:233-247) and post-processing (:255-270); never reached in this image
        destination = destination_of(filename, for_opentrack, quantize, fp16)
        print(f"Exporting {wrapped.__class__}, input size = {inputs[0].shape[2]},{inputs[0].shape[3]} to {destination}")
        torch.onnx.export(wrapped, inputs, destination, training=torch.onnx.TrainingMode.EVAL, export_params=True, opset_version=OPSET_VERSION,
                          do_constant_folding=True, keep_initializers_as_inputs=False, input_names=wrapped.input_names,
                          output_names=wrapped.output_names, dynamic_axes=dynamic_axes, verbose=False)
        import onnx.shape_inference
        onnxmodel = onnx.load(destination)
        onnxmodel.doc_string, onnxmodel.model_version = DOC_STRING, MODEL_VERSION
        onnxmodel = onnx.shape_inference.infer_shapes(onnxmodel)
        onnx.checker.check_model(onnxmodel)
        onnx.save(onnxmodel, destination)
        contract["file"] = destination
    else:
        destination = destination_of(filename, for_opentrack, quantize, fp16, ext=".pt")
        print(f"`onnx` is not installed: writing the traced graph {destination} and its contract instead of an .onnx file")
        traced.save(destination)
        contract["file"] = destination
    with open(splitext(contract["file"])[0] + ".contract.json", "w") as fh:
        json.dump(contract, fh, indent=1)
    return contract


class _ExportLike(nn.Module):
    """ExportModel over an already-probed list of output names (the folded network has the same outputs as the original)."""

    def __init__(self, original, output_names):
        super().__init__()
        self._original, self.output_names, self.input_names = original, list(output_names), ["x"]

    def forward(self, x):
        y = self._original(x)
        return tuple(_as_tensor(y[k]) for k in self.output_names)


def main():
    parser = argparse.ArgumentParser(description="Convert networks to onnx format")
    parser.add_argument("--posenet", dest="posemodelfilename", help="filename of model checkpoint", type=str, default=None)
    parser.add_argument("--full", action="store_true", default=False)
    parser.add_argument("--localizer", dest="localizermodelfilename", type=str, default=None)
    parser.add_argument("--quantize", action="store_true", default=False)
    parser.add_argument("--fp16", action="store_true", default=False)
    args = parser.parse_args()
    if args.localizermodelfilename:
        raise NotImplementedError("the face localizer network (reference :172-176, 282-324) is outside the pose-estimator path (SURVEY.md §2)")
    if args.posemodelfilename:
        net = trackertraincode.neuralnets.models.load_model(args.posemodelfilename)
        contract = convert_posemodel_onnx(net, args.posemodelfilename, for_opentrack=not args.full, quantize=args.quantize, fp16=args.fp16)
        print(json.dumps(contract, indent=1))


if __name__ == "__main__":
    main()
