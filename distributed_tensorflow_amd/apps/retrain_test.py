"""Retrained-classifier inference driver (``retrain1/test.py`` / ``retrain2/test.py``, component C40).

Reference (``retrain1/test.py:10-59``): read ``retrained_labels.txt``, import the frozen graph, and for
every file under ``imgs`` feed the raw JPEG, fetch ``final_result:0``, show the image, print every class
sorted by score and the top-1.  Here all images are decoded on host threads, preprocessed and pushed
through the batched backbone + head in one go; display is opt-in and headless-safe.
"""
from __future__ import annotations

import argparse
import os
from typing import List

import numpy as np


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Classify images with a retrained bundle")
    p.add_argument("--imgs", type=str, default="imgs")
    p.add_argument("--graph", type=str, default="retrained_graph.pb")
    p.add_argument("--labels", type=str, default="retrained_labels.txt")
    p.add_argument("--device", type=str, default="auto")
    p.add_argument("--show", action="store_true")
    p.add_argument("--backbone_precision", type=str, default="bf16", choices=["bf16", "fp32"])
    return p


def run(args) -> List[str]:
    import torch

    from ..data.bottleneck import ImagePreprocessor
    from ..data.imageprepare import walk_images
    from ..models.retrain_head import RetrainHead, TinyBackbone
    from ..utils import checkpoint as ckpt

    lines = open(args.labels).readlines()
    uid_to_human = {uid: line.strip("\n") for uid, line in enumerate(lines)}  # retrain1/test.py:10-16
    use_cuda = torch.cuda.is_available() and args.device != "cpu"
    device = torch.device("cuda" if use_cuda else "cpu")
    from ..utils import graphdef as gd
    fp32_input = False
    if gd.is_graphdef(args.graph):
        # the reference's artefact (retrain1/test.py:26-33): a serialized GraphDef holding Inception-v3 + the new layer;
        # parsed without TensorFlow, fed DecodeJpeg/contents -> final_result
        layers, fw, fb, final_name, ops = gd.read_retrained_graph(args.graph)
        from ..models.inception_v3 import InceptionV3
        net = InceptionV3(device, seed=0)
        if sum(1 for _n, o, _i in ops if o == "Conv2D") != len(net.graph.convs) or set(layers) != set(net.graph.convs):
            raise RuntimeError("%s is not the 2015 Inception-v3 topology (%d conv layers)" % (args.graph, len(layers)))
        net.load_tf_weights(layers)
        head = RetrainHead(int(fb.shape[0]), device, final_tensor_name=final_name)
        head.load_tf_tensors({"final_training_ops/weights/final_weights": fw, "final_training_ops/biases/final_biases": fb})
        backbone = net.forward
        if args.backbone_precision == "fp32" or not use_cuda:  # the graph's own precision (models/inception_v3.py Fp32Backbone); the only path without a GPU
            from ..models.inception_v3 import Fp32Backbone
            backbone, fp32_input = Fp32Backbone(net), True
    else:
        t = ckpt.load_tensors(args.graph)
        backbone_kind = bytes(t["_dtf/backbone"]).decode() if "_dtf/backbone" in t else "?"
        if backbone_kind != "tiny":
            raise RuntimeError("%s: not a GraphDef and not a stand-in-backbone bundle (written by an older build?): "
                               "re-run retrain.py" % args.graph)
        C = t["final_training_ops/biases/final_biases"].shape[0]
        head = RetrainHead(C, device)
        head.load_tf_tensors(t)
        backbone = TinyBackbone(device, seed=int(t["_dtf/backbone_seed"]))
        if "backbone/tiny/proj" in t:
            backbone.proj = torch.from_numpy(np.asarray(t["backbone/tiny/proj"], np.float32)).to(device)
    files = walk_images(args.imgs, (".jpg", ".jpeg"))
    if not files:
        raise FileNotFoundError("no JPEG images under %r" % args.imgs)
    x = ImagePreprocessor(device, fp32=fp32_input).load(files)
    probs = head.predict(backbone(x)).cpu().numpy()
    top1 = []
    for f, p in zip(files, probs):
        print(f)
        if args.show:
            try:
                import matplotlib
                if not os.environ.get("DISPLAY"):
                    matplotlib.use("Agg")
                import matplotlib.pyplot as plt
                from PIL import Image
                plt.imshow(Image.open(f))
                plt.axis("off")
                plt.show()
            except Exception as e:  # noqa: BLE001
                print("(display skipped: %r)" % (e,))
        top_k = p.argsort()[::-1]
        for node_id in top_k:
            print("%s (score = %.5f)" % (uid_to_human[int(node_id)], p[node_id]))
        print("=> " + uid_to_human[int(top_k[0])])
        print()
        top1.append(uid_to_human[int(top_k[0])])
    return top1


def main(argv=None):
    args, _ = build_parser().parse_known_args(argv)
    return run(args)
