"""Stage 2 -- generator warm-up (reference: src/main_warmup.py).

    python -m consistent__style_transfer_amd.main_warmup --dataset=yelp --ver=0

Denoising auto-encoding of the DenoiseLSTM generator (token CE), Adam(lr 1e-3), clip 1.0; the stage
forces epochs=1 and batch_size=512 exactly as main_warmup.py:115-122 does (an explicit --batch_size
wins); the best generator goes to `<dump_dir>/<dataset>/warmup/G.pth`.
"""
import os
import sys

import torch

from . import ops
from .arguments import apply_model_constants, fetch_args
from .loader import StyleDataset, collate_warmup, load_s2l
from .stages import WarmupStage
from .trainer import StepCache, Trainer
from .vocab import BPETokenizer

STAGE = "warmup"


class WarmupAdapter(WarmupStage):
    def __init__(self, args, vocab):
        super().__init__(len(vocab), args.n_class, args.max_len)
        self.hparams = args
        self.best_eval = float("inf")

    def train_batch(self, trainer, batch, batch_idx):
        coins = trainer.coins(batch[1].shape[1])
        out = self._steps.run("w", lambda nx, x, lab, c, reducer=None: self.train_step((nx, x, lab), coins=c, reducer=reducer),
                              list(batch) + [coins])
        return {"dn_loss": out["dn_loss"]}

    def validation_step(self, trainer, batch):
        coins = trainer.coins(batch[1].shape[1])
        return float(self.loss(batch, coins=coins).item())

    def validation_end(self, trainer, outputs):
        loss = trainer.mean_over_ranks(sum(outputs) / len(outputs))
        if self.best_eval > loss:
            self.best_eval = loss
            if trainer.rank == 0:
                torch.save(self.generator.state_dict(), f"{self.hparams.task_dump_dir}/G.pth")
        return loss


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = fetch_args(argv)
    apply_model_constants(args)
    ops.set_precision(args.precision)
    torch.manual_seed(args.seed)                              # identical random-init weights on every rank (then broadcast anyway)
    args.epochs = 1                                           # main_warmup.py:115-122
    if not any(a.startswith("--batch_size") for a in argv):
        args.batch_size = 512
    os.makedirs(f"{args.dump_dir}/{args.dataset}/{STAGE}", exist_ok=True)
    args.task_dump_dir = f"{args.dump_dir}/{args.dataset}/{STAGE}"
    args.log_dir = f"{args.log_dir}/{args.dataset}"
    vocab = BPETokenizer.load(f"{args.dump_dir}/{args.dataset}/{args.dataset}-vocab.json",
                              f"{args.dump_dir}/{args.dataset}/{args.dataset}-merges.txt")
    trainer = Trainer(args, patience=1, log_name=STAGE)
    stage = WarmupAdapter(args, vocab).to(trainer.device)
    stage.train()
    stage.setup_optim()
    trainer.sync_replicas(stage)
    stage._steps = StepCache(trainer.use_graph, [stage], trainer.reducer)
    data_dir = f"{args.data_dir}/{args.dataset}"
    train_ds = StyleDataset([f"{data_dir}/style.train.0", f"{data_dir}/style.train.1"], vocab, args.max_len, load_s2l, cache=args.token_cache)
    val_ds = StyleDataset([f"{data_dir}/style.dev.0", f"{data_dir}/style.dev.1"], vocab, args.max_len, load_s2l, cache=args.token_cache)
    trainer.fit(stage, train_ds, val_ds, collate_warmup, args.batch_size)
    return stage


if __name__ == "__main__":
    main()
