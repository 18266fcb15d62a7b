"""GPU: the eval-framework plugin level (procyon_amd/evaluate.py, mirror of procyon/evaluate/framework/procyon.py) driven by
stub loaders that follow the reference's loader protocol; results are checked against direct model calls and the oracle's
scoring."""
from types import SimpleNamespace

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


class _Loader(list):
    """a DataLoader stand-in: iterable of collated batches + .dataset.aaseq_type + .collate_fn"""

    def __init__(self, batches, collate_fn=None):
        super().__init__(batches)
        self.dataset = SimpleNamespace(aaseq_type="protein", query_is_sequence=False)
        self.collate_fn = collate_fn


def _batch(prot, instr, seq_slots, seq_ref, texts=(), text_slots=(), text_ref=(), target_text=None):
    return {"data": {"seq": prot, "seq_idx": None if prot is None else torch.arange(prot.shape[0]), "text": list(texts), "drug": None},
            "input": {"seq": seq_slots, "text": [list(t) for t in text_slots], "drug": None},
            "target": {"seq": None, "text": target_text, "drug": None}, "instructions": list(instr),
            "reference_indices": {"input": {"seq": seq_ref, "text": [list(t) for t in text_ref]}}}


@pytest.fixture(scope="module")
def model():
    from procyon_amd import synthetic_model as SM
    return SM.build("small", device="cuda", max_new_tokens=16)


def test_caption_eval_get_predictions(model):
    from procyon_amd import synth
    from procyon_amd.evaluate import ProcyonCaptionEval
    prot = synth.protein_tokens([50, 33], seed=2)
    b = _batch(prot, ["w1 <|protein|> w2 [ANSWER]", "w3 w4 <|protein|> [ANSWER]"], [[0], [1]], [[101], [7, 202]], text_slots=[[], []],
               text_ref=[[], []])
    from procyon.evaluate.framework.args import EvalArgs
    ev = ProcyonCaptionEval.from_model(model, {"generation_method": "beam", "num_captions": 2, "beam_group_size": 2}, EvalArgs(caption_max_len=5))
    df = ev.get_predictions(_Loader([b]))
    assert list(df.columns) == ["seq_id", "generated_caption"] and len(df) == 4
    assert df["seq_id"].tolist() == [101, 101, 202, 202]
    _, _, _, caps = model.generate(b, max_len=5, method="beam", beam_size=4, beam_group_size=2, truncate_on_eos=True)
    assert df["generated_caption"].tolist() == [caps[0][0], caps[0][2], caps[1][0], caps[1][2]]


def test_qa_eval_get_predictions(model):
    from procyon_amd import synth
    from procyon_amd.evaluate import ProcyonQAEval, get_qa_scores
    prot = synth.protein_tokens([50, 33, 20], seed=4)
    instr = ["w1 <|protein|> ok ? [ANSWER] yes w2 <|protein|> ok ? [ANSWER]", "w5 <|protein|> ok ? [ANSWER]"]
    b = _batch(prot, instr, [[0, 1], [2]], [[11, 12], [13]], text_slots=[[], []], text_ref=[[5], [6]], target_text=["yes", "no"])
    cf = SimpleNamespace(_get_input_contexts=lambda a, b_: None)
    from procyon.evaluate.framework.args import EvalArgs
    ev = ProcyonQAEval.from_model(model, {}, EvalArgs())
    res = ev.get_predictions(_Loader([b, b], collate_fn=cf))
    assert res["seq_ids"] == [12, 13, 12, 13] and res["text_ids"] == [5, 6, 5, 6]
    assert res["y"].tolist() == [model.yes_token, model.no_token] * 2
    out = model(b, retrieval=False, get_full_labels=True, crop_off=True)
    pred, _ = get_qa_scores(out, answer_token=model.answer_idx)
    assert torch.equal(res["pred"], torch.cat([pred, pred]))
    assert torch.equal(pred, out["outputs"].answer_logits[:, 0].float().argmax(-1).cpu())


def test_retrieval_eval_get_predictions(model):
    from oracle.procyon_ref import retrieval_scores
    from procyon_amd import synth
    from procyon_amd.evaluate import ProcyonRetrievalEval
    prots = synth.protein_tokens([40, 25, 61, 18, 33], seed=9)
    ids = torch.tensor([3, 0, 4, 1, 2])                       # target loader yields integer ids in its own order

    class _CF:
        def _convert_batch(self, kind, protein_ids):
            assert kind == "sequence"
            return prots[protein_ids]

    q1 = _batch(None, ["w1 w2 find [PROT]", "w3 [EXT] find [PROT]"], None, [[], []], texts=["alpha"], text_slots=[[], [0]], text_ref=[[70], [71]])
    q2 = _batch(None, ["w4 w4 find [PROT]"], None, [[]], text_slots=[[]], text_ref=[[70]])     # query 70 again: the last one wins
    q1["data"]["seq"] = None
    q2["data"]["seq"] = None
    from procyon.evaluate.framework.args import EvalArgs
    ev = ProcyonRetrievalEval.from_model(model, {}, EvalArgs(retrieval_use_cached_target_embeddings=False))
    sims = ev.get_predictions(_Loader([q1, q2], collate_fn=_CF()), [ids[:3], ids[3:]], query_order=[71, 70], target_order=[0, 1, 2, 3, 4])
    assert sims.dtype == torch.float64 and sims.shape == (2, 5) and sims.device.type == "cpu"
    qe = torch.cat([model(q1, retrieval=True)["contrastive_out"]["positive"]["text"][1:2],
                    model(q2, retrieval=True)["contrastive_out"]["positive"]["text"][0:1]]).cpu()
    te = model.forward_sequences(prots)["shared"].cpu()
    ref = retrieval_scores(qe, te)
    assert rel_err(sims, ref) < 5e-3
    assert torch.equal(sims.argmax(-1), ref.argmax(-1))


def _write_checkpoint_dir(d, model_w, tokenizer_geom=None):
    """a checkpoint directory in the reference's layout (model_args.pt / data_args.pt / txllm_model_ckpt.pt) holding the small
    synthetic model's weights under the reference's state-dict names (HF Esm names for the encoder)"""
    import os
    from procyon.training.training_args_IT import DataArgs, ModelArgs
    w = model_w
    sd = {"text_encoder.model." + k: v for k, v in w["llama"].items()}
    sd.update({"protein_seq_encoder.model." + k: v for k, v in w["esm"].items()})
    for name, key in (("token_projectors.aaseq", "aaseq"), ("aaseq_shared_projector", "shared"), ("aaseq_lm_projector", "lm")):
        for j, (wt, b) in zip((0, 3, 6), w["projs"][key]):
            sd[f"{name}.{j}.weight"], sd[f"{name}.{j}.bias"] = wt, b
    margs = ModelArgs(protein_pooling_opt="mean", ret_token_access="last", use_aaseq_embeddings=False)
    torch.save(margs, os.path.join(d, "model_args.pt"))
    torch.save(DataArgs(data_dir="/x"), os.path.join(d, "data_args.pt"))
    torch.save(sd, os.path.join(d, "txllm_model_ckpt.pt"))
    return margs


def test_registry_style_construction_and_target_cache(tmp_path, capsys, monkeypatch):
    """The registry contract (core.py:210-216): `model_zoo[task][model_type](model_config, eval_args, model_args, device)` builds each
    plugin from a checkpoint DIRECTORY; predictions equal those of a plugin wrapped around the directly built model; the retrieval
    plugin writes and then re-reads `<checkpoint_dir>/protein_target_embeddings.pkl` (procyon.py:324-355)."""
    import os
    import pandas as pd
    from procyon.evaluate.framework.args import EvalArgs
    from procyon.evaluate.framework.procyon import ProcyonCaptionEval, ProcyonQAEval, ProcyonRetrievalEval
    from procyon.training.training_args_IT import ModelArgs
    from procyon_amd import synth
    from procyon_amd import synthetic_model as SM
    model, w = SM.build("small", device="cuda", max_new_tokens=16, return_weights=True)
    ckpt = str(tmp_path / "ckpt")
    os.makedirs(ckpt)
    _write_checkpoint_dir(ckpt, w)
    model_zoo = {"caption": {"ProCyon": ProcyonCaptionEval}, "qa": {"ProCyon": ProcyonQAEval}, "retrieval": {"ProCyon": ProcyonRetrievalEval}}
    args = {"model_type": "ProCyon", "checkpoint_dir": ckpt, "num_captions": 2, "beam_group_size": 2, "tokenizer": model.tokenizer,
            "engine_kwargs": dict(max_new_tokens=16, esm_heads=2, head_dim=64, max_pos=4096)}
    eval_args = EvalArgs(caption_max_len=4, batch_size=2)
    device = torch.device("cuda")
    # a deliberately different ModelArgs: the mismatch warning of compare_and_warn_model_args must name the field
    cap = model_zoo["caption"][args["model_type"]](args, eval_args, ModelArgs(protein_pooling_opt="max", ret_token_access="last", use_aaseq_embeddings=False), device)
    assert "protein_pooling_opt: max != mean" in capsys.readouterr().out
    assert cap.model.dtype == torch.bfloat16 and cap.max_len == 4 and cap.beam_size == 4 and cap.checkpoint_dir == ckpt
    prot = synth.protein_tokens([50, 33], seed=2)
    b = _batch(prot, ["w1 <|protein|> w2 [ANSWER]", "w3 w4 <|protein|> [ANSWER]"], [[0], [1]], [[101], [7, 202]], text_slots=[[], []], text_ref=[[], []])
    df = cap.get_predictions(_Loader([b]))
    df_ref = ProcyonCaptionEval.from_model(model, args, eval_args).get_predictions(_Loader([b]))
    assert df.equals(df_ref)
    qa = model_zoo["qa"]["ProCyon"](args, eval_args, None, device)
    assert (qa.yes_token, qa.no_token) == (model.yes_token, model.no_token)
    # retrieval with the on-disk cache: the entity table lives under $DATA_DIR, ids are rows of a token matrix
    prots = synth.protein_tokens([40, 25, 61, 18, 33], seed=9)
    data_dir = tmp_path / "data"
    os.makedirs(data_dir / "integrated_data/v1/protein")
    pd.DataFrame({"name": list("abcde")}, index=[0, 1, 2, 3, 4]).to_pickle(data_dir / "integrated_data/v1/protein/protein_info_filtered.pkl")
    monkeypatch.setenv("DATA_DIR", str(data_dir))

    class _CF:
        def _convert_batch(self, kind, protein_ids):
            assert kind == "sequence"
            return prots[protein_ids]

    ret = model_zoo["retrieval"]["ProCyon"](args, eval_args, None, device)
    assert ret.use_cached_target_embeddings and ret.batch_size == 2
    q1 = _batch(None, ["w1 w2 find [PROT]", "w3 w9 find [PROT]"], None, [[], []], text_slots=[[], []], text_ref=[[70], [71]])
    q1["data"]["seq"] = None
    mk = lambda: _Loader([{**q1, "target": dict(q1["target"])}], collate_fn=_CF())
    cache = os.path.join(ckpt, "protein_target_embeddings.pkl")
    assert not os.path.exists(cache)
    sims = ret.get_predictions(mk(), None, query_order=[71, 70], target_order=[4, 2, 0])
    assert os.path.exists(cache) and sims.shape == (2, 3) and sims.dtype == torch.float64
    emb, ids = torch.load(cache, weights_only=False)
    assert ids == [0, 1, 2, 3, 4] and emb.device.type == "cpu" and emb.shape[0] == 5
    assert torch.equal(emb.cuda(), model.forward_sequences(prots)["shared"])
    # second call reads the file: poison it and see the poison
    torch.save((torch.zeros_like(emb), ids), cache)
    sims2 = ret.get_predictions(mk(), None, query_order=[71, 70], target_order=[4, 2, 0])
    assert float(sims2.abs().max()) == 0.0
    # ... and the cache-less path agrees with the first answer
    ret.use_cached_target_embeddings = False
    ids_t = torch.tensor([0, 1, 2, 3, 4])
    sims3 = ret.get_predictions(mk(), [ids_t[:2], ids_t[2:]], query_order=[71, 70], target_order=[4, 2, 0])
    assert torch.equal(sims3, sims)
