"""QA filtering of generated captions over the MI355X engine: the command line of the reference's scripts/qa_filter_captions.py,
batched across (protein, caption) pairs (`--batch_size`).  See procyon_amd/pipelines.py.

Arithmetic: the reference script runs the model as loaded -- fp32 (/root/reference/scripts/qa_filter_captions.py:17-18, no
`.bfloat16()`) -- and so does this one by default (the fp32 operator family, procyon_amd/engine_f32.py); `--bf16` casts the model first,
as the evaluation framework does, for ~10x the rate."""
import argparse

import torch

from procyon.model.model_unified import UnifiedProCyon
from procyon_amd.pipelines import chunk_rows, load_caption_table, qa_filter_captions


def main(args):
    device = torch.device("cuda")
    data_args, _, _ = UnifiedProCyon.get_checkpoint_configs(resume_from_checkpoint=args.ckpt)
    model, _ = UnifiedProCyon.from_pretrained(checkpoint_dir=args.ckpt)
    if args.bf16:
        model.bfloat16()
    model.to(device)
    model.eval()
    captions = load_caption_table(args.caption_fpath, args.caption_dir)
    start, end = chunk_rows(captions.shape[0], args.num_chunks, args.chunk_idx, "qa_filter_captions")
    print(qa_filter_captions(model, data_args, captions.iloc[start:end, :], args.prompt_dataset, args.prompt_relation, args.save_path,
                             args.batch_size, device=device))


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--ckpt", required=True, help="Path to ProCyon checkpoint")
    p.add_argument("--caption_dir", type=str, help="Directory with captions")
    p.add_argument("--caption_fpath", type=str, help="File with captions to filter, in the style of caption_bulk.py output")
    p.add_argument("--save_path", default=None, type=str, help="CSV file to save to")
    p.add_argument("--chunk_idx", default=None, type=int)
    p.add_argument("--num_chunks", default=None, type=int)
    p.add_argument("--prompt_dataset", default="uniprot")
    p.add_argument("--prompt_relation", default="all")
    p.add_argument("--batch_size", default=16, type=int, help="pairs per engine call (the reference script: 1)")
    p.add_argument("--bf16", action="store_true", help="cast the model to bfloat16 first (the reference script scores in fp32)")
    a = p.parse_args()
    assert (a.caption_fpath is not None) or (a.caption_dir is not None)
    main(a)
