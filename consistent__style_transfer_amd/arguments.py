"""Command line of the three stage scripts (reference: src/arguments.py).

Every reference flag keeps its name, type and default; `--dataset` still derives `max_len` /
`batch_size` (yelp: 18 / 256, book: 30 / 128, anything else: ValueError).  Additive flags expose what
the reference hard-codes as module constants and the new run-time
choices (precision, graphs, data parallelism); with none of them given the behaviour is the
reference's.
"""
import argparse

base_dir = ".."


def build_parser():
    parser = argparse.ArgumentParser(description="Parameters")
    parser.add_argument('--dataset', type=str, required=True)
    parser.add_argument('--mode', type=str, default="train")            # "train" or "test"
    parser.add_argument('--ver', type=str, required=True)
    # file system
    parser.add_argument('--data_dir', type=str, default=f"{base_dir}/data")
    parser.add_argument('--dump_dir', type=str, default=f"{base_dir}/dump")
    parser.add_argument('--log_dir', type=str, default=f"{base_dir}/log")
    parser.add_argument('--out_dir', type=str, default=f"{base_dir}/output")
    # model setting
    parser.add_argument('--n_class', type=int, default=2, help="number of styles")
    parser.add_argument('--p_drop', type=float, default=0.1, help="dropout rate (parsed, unused: as in the reference)")
    parser.add_argument('--w_s', type=float, default=0.1, help="weight of STI")
    parser.add_argument('--w_c', type=float, default=0.5, help="weight of CP")
    parser.add_argument('--w_adv', type=float, default=1.0, help="weight of adversarial loss")
    parser.add_argument('--w_bt', type=float, default=1.0, help="without back-trans")
    parser.add_argument('--tau', type=float, default=0.1, help="annealling temperature")
    parser.add_argument('--gap', type=float, default=0.0, help="annealling temperature")
    parser.add_argument('--epochs', type=int, default=10, help="max epochs")
    parser.add_argument('--device', type=str, default="0", help="device id")
    parser.add_argument('--restore_version', type=int, default=-1, help="version for restore trainer and it's state")
    # ---- additive flags (not in the reference) ----------------------------------------------
    parser.add_argument('--batch_size', type=int, default=None, help="override the dataset-derived (global) batch size")
    parser.add_argument('--max_len', type=int, default=None, help="override the dataset-derived max length")
    parser.add_argument('--n_layer', type=int, default=None, help="encoder layers of MLM / Matcher (reference constant 6)")
    parser.add_argument('--d_model', type=int, default=None, help="width of MLM / Matcher (reference constant 512)")
    parser.add_argument('--n_head', type=int, default=None, help="attention heads (reference constant 8)")
    parser.add_argument('--precision', type=str, default="bf16", choices=["bf16", "f32", "fp8w"],
                        help="MFMA arithmetic of the GEMMs: bf16 operands / fp32 accumulate, exact fp32, or bf16 with fp8 (e4m3, "
                             "per-output-channel scale) weights in the encoder layers' QKV / out-projection / FFN products")
    parser.add_argument('--no_graph', action="store_true", help="launch kernels eagerly instead of replaying hipGraphs")
    parser.add_argument('--seed', type=int, default=0, help="base seed (data order, noise, dropout, coins)")
    parser.add_argument('--max_steps', type=int, default=None, help="stop after this many training batches (smoke runs)")
    parser.add_argument('--val_batches', type=int, default=None, help="limit validation batches")
    parser.add_argument('--replica_check_every', type=int, default=200,
                        help="data parallel: verify every N steps that all ranks hold bit-identical parameters (0 = only per epoch)")
    parser.add_argument('--token_cache', action="store_true",
                        help="keep / reuse the binary token cache next to each data file (loader.TokenCache)")
    parser.add_argument('--label_cache', type=str, default=None,
                        help="pretrain: file of precomputed content-distance labels (loader.LabelCache format)")
    parser.add_argument('--prefetch_workers', type=int, default=0,
                        help="build batches (token noise, padding, WMD labels) in this many worker processes ahead of the GPU")
    return parser


def finish_args(args):
    if args.dataset == "yelp":
        max_len, batch_size = 18, 256
    elif args.dataset == "book":
        max_len, batch_size = 30, 128
    else:
        raise ValueError
    if getattr(args, "max_len", None) is None:
        args.max_len = max_len
    if getattr(args, "batch_size", None) is None:
        args.batch_size = batch_size
    return args


def fetch_args(argv=None):
    return finish_args(build_parser().parse_args(argv))


def apply_model_constants(args):
    """The reference's sizes are module-level constants read at construction;
    the additive flags override them the same way."""
    from .model import match, mlm
    for name in ("n_layer", "d_model", "n_head"):
        v = getattr(args, name, None)
        if v is not None:
            setattr(mlm, name, v)
            setattr(match, name, v)
