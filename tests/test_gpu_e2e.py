"""GPU end to end: the three stage scripts behind the reference's CLI on a small corpus (the Yelp
dev sample fixture): pretrain -> warmup -> optimize (train) -> optimize (test), checkpoints under the
reference's file names, state_dict keys loadable across stages, .tsf writers, hipGraph and eager."""
import json
import os
import shutil

import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _small_constants():
    from consistent__style_transfer_amd.model import classifier, discriminator, rnn
    rnn.d_embed, rnn.d_enc, rnn.d_dec = 32, 32, 64
    classifier.d_embed, classifier.kernel_number = 32, [16, 16, 16]
    discriminator.embed_dim, discriminator.num_rep, discriminator.dis_num_filters = 32, 4, [12, 12, 12, 12]


def _restore_constants():
    from consistent__style_transfer_amd.model import classifier, discriminator, match, mlm, rnn
    rnn.d_embed, rnn.d_enc, rnn.d_dec = 128, 256, 512
    classifier.d_embed, classifier.kernel_number = 128, [128, 128, 128]
    discriminator.embed_dim, discriminator.num_rep, discriminator.dis_num_filters = 128, 16, [300, 300, 300, 300]
    mlm.d_model = match.d_model = 512
    mlm.n_head = match.n_head = 8
    mlm.n_layer = match.n_layer = 6


@pytest.mark.parametrize("graph", [True, False])
def test_three_stages_cli(tmp_path, graph):
    from consistent__style_transfer_amd import main_optimize, main_pretrain, main_warmup
    root = tmp_path
    data, dump = root / "data" / "yelp", root / "dump" / "yelp"
    os.makedirs(data)
    os.makedirs(dump)
    for lab in (0, 1):
        src = os.path.join(G, f"yelp_dev_sample.{lab}")
        for split in ("train", "dev", "test"):
            shutil.copy(src, data / f"style.{split}.{lab}")
    shutil.copy(os.path.join(G, "yelp_sample-vocab.json"), dump / "yelp-vocab.json")
    shutil.copy(os.path.join(G, "yelp_sample-merges.txt"), dump / "yelp-merges.txt")
    common = ["--dataset", "yelp", "--data_dir", str(root / "data"), "--dump_dir", str(root / "dump"),
              "--log_dir", str(root / "log"), "--out_dir", str(root / "output"), "--n_layer", "1", "--d_model", "64",
              "--n_head", "4", "--batch_size", "32", "--max_steps", "6", "--val_batches", "2", "--epochs", "1"]
    if not graph:
        common.append("--no_graph")
    else:
        common += ["--token_cache", "--prefetch_workers", "2"]      # binary token cache + batches built ahead in workers
    _small_constants()
    try:
        if graph:
            # word vectors in this build's container format next to the vocabulary: main_pretrain then takes the REAL label path
            # (wmd.WMDdistance -> libcst_host.so cst_host_wmd_labels) instead of the overlap stand-in (round-2 verdict, f1)
            from consistent__style_transfer_amd import wmd as _wmd
            from consistent__style_transfer_amd.vocab import BPETokenizer
            vocab = BPETokenizer.load(str(dump / "yelp-vocab.json"), str(dump / "yelp-merges.txt"))
            _wmd.WMDdistance.train([str(data / "style.train.0"), str(data / "style.train.1")], vocab, dim=8).save(str(dump / "yelp-w2v.npz"))
            calls = []
            orig = _wmd.WMDdistance.cal_wmd_label

            def spy(self, xs1, xs2, tok, rows=None, nthreads=1):
                out = orig(self, xs1, xs2, tok, rows=rows, nthreads=nthreads)
                calls.append((len(xs1), float(max(out)), float(min(out))))
                return out
            _wmd.WMDdistance.cal_wmd_label = spy
        pre = main_pretrain.main(common + ["--ver", "0"] + (["--prefetch_workers", "0"] if graph else []))
        if graph:
            _wmd.WMDdistance.cal_wmd_label = orig
            assert len(calls) >= 6 and all(n == 32 for n, _, _ in calls[:3]) and max(hi for _, hi, _ in calls) > 0.05, calls[:4]
        for name in ("cls", "mat", "dn"):
            assert os.path.exists(dump / "pretrain" / f"{name}.pth")
        assert all(torch.isfinite(p).all() for p in pre.parameters())
        wu = main_warmup.main(common + ["--ver", "0"])
        assert os.path.exists(dump / "warmup" / "G.pth")
        assert wu.best_eval < 10.0
        opt = main_optimize.main(common + ["--ver", "v0"])
        saved = os.listdir(dump / "optimize-v0")
        assert len(saved) == 1 and saved[0].startswith("G_epoch_")
        assert all(torch.isfinite(p).all() for p in opt.generator.parameters())
        from consistent__style_transfer_amd import ops as _ops
        main_optimize.main(common + ["--ver", "v0", "--mode", "test"])
        assert _ops.get_precision() == "f32"                       # bulk transfer defaults to the mode whose ids are pinned
        _ops.set_precision("bf16")
        for split in ("train", "test"):
            for lab in (0, 1):
                lines = open(root / "output" / "yelp-v0" / f"style.{split}.{lab}.tsf", encoding="utf-8").read().split("\n")
                assert len(lines) - 1 == 150
            assert not [f for f in os.listdir(root / "output" / "yelp-v0") if ".part" in f]
        if graph:
            assert os.path.exists(str(data / "style.train.0") + ".tok18.cstt")
        # scalar stream with the reference's names
        log = root / "log" / "yelp" / "optimize-v0"
        ver = sorted(os.listdir(log))[0]
        rows = [json.loads(l) for l in open(log / ver / "metrics.jsonl")]
        assert any("val_loss" in r for r in rows)
        assert os.path.exists(log / ver / "meta_tags.csv")
    finally:
        _restore_constants()


# ------------------------------------------------------------------------------ data-parallel transfer writer (main_optimize --mode test)
def _prepare_transfer_dirs(root):
    """Corpus + tokenizer + random-init checkpoints under the reference's file names: enough for `--mode test`."""
    from consistent__style_transfer_amd.model import MLM, DenoiseLSTM, Matcher, TextCNN
    from consistent__style_transfer_amd.vocab import BPETokenizer
    data, dump = root / "data" / "yelp", root / "dump" / "yelp"
    os.makedirs(data)
    os.makedirs(dump / "pretrain")
    os.makedirs(dump / "warmup")
    for lab in (0, 1):
        src = os.path.join(G, f"yelp_dev_sample.{lab}")
        lines = open(src, encoding="utf-8").read().split("\n")[:37 + 6 * lab]          # 37 + 43 = 80 sentences: not a multiple of 32
        for split in ("train", "test"):
            with open(data / f"style.{split}.{lab}", "w", encoding="utf-8") as f:
                f.write("\n".join(lines) + "\n")
    shutil.copy(os.path.join(G, "yelp_sample-vocab.json"), dump / "yelp-vocab.json")
    shutil.copy(os.path.join(G, "yelp_sample-merges.txt"), dump / "yelp-merges.txt")
    vocab = BPETokenizer.load(str(dump / "yelp-vocab.json"), str(dump / "yelp-merges.txt"))
    from consistent__style_transfer_amd.model import match, mlm
    mlm.d_model = match.d_model = 64                           # what --d_model / --n_head / --n_layer below will set (arguments.apply_model_constants)
    mlm.n_head = match.n_head = 4
    mlm.n_layer = match.n_layer = 1
    torch.manual_seed(5)
    V = len(vocab)
    torch.save(TextCNN(V, n_class=2).state_dict(), dump / "pretrain" / "cls.pth")
    torch.save(Matcher(V).state_dict(), dump / "pretrain" / "mat.pth")
    torch.save(MLM(V, 2).state_dict(), dump / "pretrain" / "dn.pth")
    torch.save(DenoiseLSTM(V, 2, 18).state_dict(), dump / "warmup" / "G.pth")


def _transfer_args(root, out):
    return ["--dataset", "yelp", "--data_dir", str(root / "data"), "--dump_dir", str(root / "dump"), "--log_dir", str(root / "log"),
            "--out_dir", str(root / out), "--n_layer", "1", "--d_model", "64", "--n_head", "4", "--batch_size", "32", "--ver", "v0",
            "--mode", "test"]


def _transfer_worker(rank, world, port, root):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      CST_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from consistent__style_transfer_amd import main_optimize
    _small_constants()
    try:
        main_optimize.main(_transfer_args(root, "out2"))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_transfer_writer_two_ranks_equals_one(tmp_path):
    """main_optimize.py:157-174 under data parallelism: two ranks (gloo, one GPU) decode their halves of every global batch and rank 0
    merges the part files -- the same sentences in the same order as the one-process run, nothing dropped (80 sentences per split,
    batches of 32: the last batch is short and odd-sized shards get padded), no part files left behind."""
    import socket

    import torch.multiprocessing as mp
    from consistent__style_transfer_amd import main_optimize
    from consistent__style_transfer_amd import ops as _ops
    root = tmp_path
    _small_constants()
    try:
        _prepare_transfer_dirs(root)
        main_optimize.main(_transfer_args(root, "out1"))
        _ops.set_precision("bf16")
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=_transfer_worker, args=(r, 2, port, root)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0
        for split in ("train", "test"):
            for lab, n in ((0, 37), (1, 43)):
                a = open(root / "out1" / "yelp-v0" / f"style.{split}.{lab}.tsf", encoding="utf-8").read().split("\n")
                b = open(root / "out2" / "yelp-v0" / f"style.{split}.{lab}.tsf", encoding="utf-8").read().split("\n")
                assert len(a) - 1 == n and len(b) - 1 == n
                same = sum(x == y for x, y in zip(a[:n], b[:n]))
                assert same == n, (split, lab, same, n)            # exact-fp32 ids: the exact mode's K slices depend on (N, K) only, never on the shard's rows
        assert not [f for f in os.listdir(root / "out2" / "yelp-v0") if ".part" in f]
    finally:
        _restore_constants()
