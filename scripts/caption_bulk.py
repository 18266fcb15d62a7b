"""Bulk captioning over the MI355X engine: the command line of the reference's scripts/caption_bulk.py (same arguments, same output
files), batched across proteins (`--batch_size`).  See procyon_amd/pipelines.py.

Like the reference script it generates with the model AS LOADED (fp32 checkpoint -> fp32 arithmetic, /root/reference/scripts/caption_bulk.py:70-73):
the fp32 operator family with its cached decode step -- a compatibility path, one launch per operator.  `--bf16` casts the model first, as the
evaluation framework and the retrieval service do (evaluate/framework/procyon.py:64-65): the fused bf16 engine (one launch per decode step),
~10 x faster."""
import argparse

import pandas as pd
import torch

from procyon.model.model_unified import UnifiedProCyon
from procyon.training.train_utils import set_seed
from procyon_amd.pipelines import caption_bulk, chunk_rows


def main(args):
    device = torch.device("cuda")
    data_args, model_args, _ = UnifiedProCyon.get_checkpoint_configs(resume_from_checkpoint=args.ckpt)
    model, _ = UnifiedProCyon.from_pretrained(checkpoint_dir=args.ckpt)
    if args.bf16:
        model.bfloat16()
    else:
        print("caption_bulk: running in fp32 like the reference script (one launch per operator, a host round trip per step: ~10 x slower than "
              "--bf16, the fused engine)", flush=True)
    model.to(device)
    model.eval()
    set_seed(1234)
    table = pd.read_csv(args.uniprot_id_file)
    start, end = chunk_rows(table.shape[0], args.num_chunks, args.chunk_idx, "caption_bulk")
    df = caption_bulk(model, model_args, data_args, table["uniprot_id"].iloc[start:end].tolist(), args.prompt_dataset, args.prompt_relation,
                      args.max_len, args.beam_size, args.diversity_penalty, args.save_path, args.batch_size, device=device)
    print(df)


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--ckpt", required=True, help="Path to ProCyon checkpoint")
    p.add_argument("--save_path", default=None, type=str, help="CSV file name to save captions to")
    p.add_argument("--chunk_idx", default=None, type=int)
    p.add_argument("--num_chunks", default=None, type=int)
    p.add_argument("--uniprot_id_file", required=True, type=str, help="CSV with a uniprot_id column")
    p.add_argument("--prompt_dataset", default="uniprot")
    p.add_argument("--prompt_relation", default="all")
    p.add_argument("--beam_size", default=10, type=int)
    p.add_argument("--max_len", default=200, type=int)
    p.add_argument("--diversity_penalty", default=0.8, type=float)
    p.add_argument("--bf16", action="store_true", help="cast the model to bfloat16 first (the fused engine; the reference script runs fp32)")
    p.add_argument("--batch_size", default=8, type=int, help="proteins per engine call (the reference script: 1)")
    main(p.parse_args())
