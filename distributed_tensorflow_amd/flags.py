"""Flag system (component C42).

The reference scripts use ``argparse`` + ``parse_known_args()`` into a module-global
``FLAGS`` (``demo2/train.py:196-223``, ``retrain1/retrain.py:479-633``,
``retrain2/retrain2.py:511-683``).  Every reference flag keeps its name, type and
default here; framework-specific flags (device, dtype, seed, sync/async, ...) are
additive.  ``demo1`` has no flags in the reference (constants in code,
``demo1/train.py:9,132,153,156,158``) — they are exposed here as flags whose
defaults are exactly those constants.
"""
from __future__ import annotations

import argparse
from typing import List, Optional, Sequence, Tuple


def _str2bool(v) -> bool:
    if isinstance(v, bool):
        return v
    s = str(v).strip().lower()
    if s in ("1", "true", "t", "yes", "y", "on"):
        return True
    if s in ("0", "false", "f", "no", "n", "off"):
        return False
    raise argparse.ArgumentTypeError("boolean value expected, got %r" % (v,))


# --------------------------------------------------------------------------- cluster
def add_cluster_flags(p: argparse.ArgumentParser, *, ps_default: str, worker_default: str,
                      job_default: str) -> None:
    """The four cluster flags of demo2/retrain2 (``demo2/train.py:198-221``)."""
    p.register("type", "bool", _str2bool)
    p.add_argument("--ps_hosts", type=str, default=ps_default,
                   help="Comma-separated list of hostname:port pairs (parameter servers). "
                        "Kept for CLI compatibility: the first entry hosts the rendezvous store.")
    p.add_argument("--worker_hosts", type=str, default=worker_default,
                   help="Comma-separated list of hostname:port pairs; its length is the world size.")
    p.add_argument("--job_name", type=str, default=job_default, help="One of 'ps', 'worker'.")
    p.add_argument("--task_index", type=int, default=0, help="Index of task within the job (== rank).")


def add_framework_flags(p: argparse.ArgumentParser) -> None:
    """Flags that exist only in this framework (SURVEY §5.6 'New framework')."""
    p.register("type", "bool", _str2bool)
    p.add_argument("--device", type=str, default="auto", help="auto | cuda | cpu")
    p.add_argument("--backend", type=str, default="auto",
                   help="collective backend: auto | nccl (RCCL over xGMI) | gloo | local")
    p.add_argument("--num_gpus", type=int, default=0,
                   help="spawn this many local ranks (0 = use the cluster flags / torchrun env)")
    p.add_argument("--dtype", type=str, default="fp32", choices=["fp32", "bf16"],
                   help="compute precision of the native MNIST engine: fp32 = the reference's (fp32 activations, exact-fp32 "
                        "MFMA); bf16 = bf16 MFMA operands with fp32 accumulation and master weights")
    p.add_argument("--seed", type=int, default=None, help="RNG seed (reference is unseeded)")
    p.add_argument("--synthetic", type="bool", nargs="?", const=True, default=None,
                   help="force synthetic data of the reference shapes (default: auto when blobs are missing)")
    p.add_argument("--double_softmax", type="bool", nargs="?", const=True, default=True,
                   help="reproduce the reference's softmax-applied-twice loss (demo1/train.py:123,127)")
    p.add_argument("--sync", dest="sync", action="store_true", default=True,
                   help="synchronous data parallelism (all-reduce every step) [default]")
    p.add_argument("--async", dest="sync", action="store_false",
                   help="emulate the reference's asynchronous updates: local steps, parameter "
                        "averaging every --async_period steps")
    p.add_argument("--async_period", type=int, default=8)
    p.add_argument("--comm", type=str, default="auto", choices=["auto", "ipc", "rccl"],
                   help="gradient exchange between GPUs of one node: direct IPC/xGMI exchange (its step schedule is "
                        "picked by a start-up trial; DTF_DP_SCHEDULE pins one), the process group's all-reduce, or "
                        "auto = whichever a start-up trial measures as fastest")
    p.add_argument("--engine", type=str, default="auto",
                   help="auto | native (HIP fused step executor) | eager (PyTorch oracle)")
    p.add_argument("--use_graph", type="bool", nargs="?", const=True, default=True,
                   help="replay the train step as a hipGraph")
    p.add_argument("--save_model_secs", type=int, default=600,
                   help="Supervisor autosave period (demo2/train.py:172)")
    p.add_argument("--fault_inject_step", type=int, default=-1,
                   help="raise inside the training loop at this step (failure-recovery tests)")
    p.add_argument("--check_replicas", type="bool", nargs="?", const=True, default=False,
                   help="debug: verify after every eval interval that all replicas hold identical parameters")
    p.add_argument("--profile", type="bool", nargs="?", const=True, default=False,
                   help="emit roctx ranges / step timing breakdown")


# --------------------------------------------------------------------------- demo1/2
def build_mnist_parser(distributed: bool) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="MNIST CNN training (demo1/demo2)")
    p.register("type", "bool", _str2bool)
    if distributed:
        # defaults: demo2/train.py:201,207,213,219 (note the space after the comma, Q10)
        add_cluster_flags(p, ps_default="192.168.1.106:2222",
                          worker_default="192.168.1.104:2223, 192.168.1.105:2224",
                          job_default="ps")
    # reference constants exposed as flags
    p.add_argument("--data_dir", type=str, default="MNIST_data")        # demo1/train.py:6
    p.add_argument("--batch_size", type=int, default=100)               # demo1/train.py:9
    p.add_argument("--learning_rate", type=float, default=1e-4)         # demo1/train.py:132
    p.add_argument("--training_steps", type=int, default=10000)         # demo1/train.py:153
    p.add_argument("--keep_prob", type=float, default=0.7)              # demo1/train.py:156
    p.add_argument("--eval_step_interval", type=int, default=100)       # demo1/train.py:158
    p.add_argument("--log_dir", type=str, default="./logs")             # demo1/train.py:151
    p.add_argument("--checkpoint_path", type=str, default="model/train.ckpt")  # demo1/train.py:165
    p.add_argument("--eval_train_set", type="bool", nargs="?", const=True, default=True,
                   help="also evaluate on the whole training set every eval interval (demo1/train.py:161)")
    p.add_argument("--variable_summaries", type="bool", nargs="?", const=True, default=False,
                   help="opt-in per-variable mean/stddev/max/min/histogram (defined-but-unused in the reference, Q3)")
    add_framework_flags(p)
    return p


# --------------------------------------------------------------------------- retrain
def build_retrain_parser(distributed: bool) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Inception-v3 transfer learning (retrain1/retrain2)")
    p.register("type", "bool", _str2bool)
    if distributed:
        # retrain2/retrain2.py:513-536
        add_cluster_flags(p, ps_default="192.168.1.105:2222",
                          worker_default="192.168.1.104:2223, 192.168.1.106:2224",
                          job_default="worker")
    p.add_argument("--image_dir", type=str, default="./data", help="Path to folders of labeled images.")
    p.add_argument("--output_graph", type=str, default="./retrained_graph.pb",
                   help="Where to save the trained inference bundle.")
    p.add_argument("--output_labels", type=str, default="./retrained_labels.txt",
                   help="Where to save the trained graph's labels.")
    p.add_argument("--summaries_dir", type=str, default="./retrain_logs",
                   help="Where to save summary logs for TensorBoard.")
    p.add_argument("--training_steps", type=int, default=2000 if distributed else 10000,
                   help="How many training steps to run before ending.")
    p.add_argument("--learning_rate", type=float, default=0.01)
    p.add_argument("--testing_percentage", type=int, default=10)
    p.add_argument("--validation_percentage", type=int, default=10)
    # retrain2 hard-codes 10 and drops the flag (retrain2/retrain2.py:473); accepted in both here (Q14)
    p.add_argument("--eval_step_interval", type=int, default=10)
    p.add_argument("--train_batch_size", type=int, default=100)
    p.add_argument("--test_batch_size", type=int, default=-1,
                   help="-1 = the whole testing split (retrain1/retrain.py:541-551)")
    p.add_argument("--validation_batch_size", type=int, default=100)
    p.add_argument("--print_misclassified_test_images", default=False, action="store_true",
                   help="print the file names of misclassified test images (parsed but unused in the reference, Q13)")
    p.add_argument("--model_dir", type=str, default="./inception_model")
    p.add_argument("--bottleneck_dir", type=str, default="./bottlenecks",
                   help="optional on-disk mirror of the HBM-resident bottleneck cache")
    p.add_argument("--final_tensor_name", type=str, default="final_result")
    p.add_argument("--flip_left_right", default=False, action="store_true")
    p.add_argument("--random_crop", type=int, default=0)
    p.add_argument("--random_scale", type=int, default=0)
    p.add_argument("--random_brightness", type=int, default=0)
    # framework additions
    p.add_argument("--bottleneck_txt_cache", type="bool", nargs="?", const=True, default=False,
                   help="also read/write reference-compatible <image>.txt bottleneck files")
    p.add_argument("--extract_batch_size", type=int, default=64,
                   help="images per batched Inception-v3 forward during bottleneck extraction")
    p.add_argument("--decode_draft", type="bool", nargs="?", const=True, default=False,
                   help="decode JPEGs of at least twice the network's input size at reduced scale (libjpeg DCT-domain "
                        "down-scaling, PIL draft): faster extraction, but NOT the reference recipe (full decode + bilinear "
                        "resize inside the graph, retrain1/retrain.py:34-35); off by default")
    p.add_argument("--step_histograms", type="bool", nargs="?", const=True, default=False,
                   help="write the variable summaries and the pre_activations / activations histograms of the new layer "
                        "on EVERY step for the train and the validation batch like the reference's `merged` "
                        "(retrain1/retrain.py:249-258,271-279,440-446), through a device ring; default: at logging points")
    # full fine-tune mode (BASELINE config 5): train every Inception-v3 layer instead of the head only
    p.add_argument("--finetune", type="bool", nargs="?", const=True, default=False,
                   help="Fine-tune the whole network (training-mode batch norm, all conv gradients).")
    p.add_argument("--finetune_batch_size", type=int, default=32, help="Per-GPU batch of the fine-tune step.")
    p.add_argument("--finetune_optimizer", type=str, default="sgd", choices=["sgd", "adam"])
    p.add_argument("--bucket_mb", type=float, default=25.0, help="Gradient all-reduce bucket size (fine-tune DP).")
    p.add_argument("--finetune_bn_recalibrate", type=int, default=8,
                   help="re-estimate the batch-norm moving statistics over this many training batches before the export "
                        "(0 = keep the running averages, decay 0.9997: right for long runs only)")
    p.add_argument("--finetune_graph", type="bool", nargs="?", const=True, default=False,
                   help="replay pairs of fine-tune steps (sampling, forward, backward, exchange, optimizer) as one hipGraph "
                        "wherever the host has nothing to do in between.  Off by default: the step is GPU-bound either way, "
                        "and the replay of a captured multi-stream step loses most of what the concurrent streams gain "
                        "(batch 128: 6 564 images/s eager, 6 080 as graph replays, 5 920 on one stream)")
    p.add_argument("--finetune_grad_dtype", type=str, default="fp32", choices=["fp32", "bf16"],
                   help="wire format of the fine-tune's gradient buckets (bf16: half the xGMI bytes; the optimizer keeps "
                        "fp32 master weights either way)")
    p.add_argument("--log_dir", type=str, default="./logs",
                   help="Supervisor logdir of retrain2 (retrain2/retrain2.py:424: ./logs/model.ckpt-<global_step>)")
    p.add_argument("--synthetic_classes", type=int, default=5)
    p.add_argument("--synthetic_images_per_class", type=int, default=40)
    add_framework_flags(p)
    return p


def parse_host_list(spec: str) -> List[str]:
    """Split a ``--ps_hosts`` / ``--worker_hosts`` value; entries are stripped (Q10)."""
    return [h.strip() for h in spec.split(",") if h.strip()]


def parse_known(parser: argparse.ArgumentParser, argv: Optional[Sequence[str]] = None
                ) -> Tuple[argparse.Namespace, List[str]]:
    """``FLAGS, unparsed = parser.parse_known_args()`` like the reference (demo2/train.py:222)."""
    return parser.parse_known_args(argv)
