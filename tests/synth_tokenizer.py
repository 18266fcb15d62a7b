"""A small Llama-3-STYLE tokenizer built in the container (no tokenizer files ship with the image): byte-level BPE trained on a
fixed corpus with the `tokenizers` library, `<|begin_of_text|>` / `<|end_of_text|>` specials, a BOS-prepending post-processor, no
pad and no sep token -- saved as a `PreTrainedTokenizerFast` directory that `transformers.AutoTokenizer.from_pretrained` loads the
way the reference loads $LLAMA3_PATH (model_unified.py:1088-1093).  Shared by tests/golden/make_golden.py (g14: the REFERENCE's
`_init_tokenizer` over it) and the CPU test (the shim's `hf_tokenizer` over an identical directory); training is deterministic."""
import os

CORPUS = [
    "Definition: You will be shown a protein. Your job is to describe the function of the protein.",
    "Protein: Output: yes no Yes No the answer is yes . the answer is no .",
    "This protein binds DNA and regulates transcription of target genes in the nucleus.",
    "Mutations in this gene cause a rare autosomal recessive disorder of the nervous system.",
    "Drug: an inhibitor of the kinase domain with a well characterised mechanism of action.",
    "Positive example: Negative example: Description: Context: function of protein phenotype disease pathway",
    "alpha beta gamma delta epsilon zeta eta theta iota kappa lambda mu filler tail words only one description",
] * 4


def build(path, vocab_size=420):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors, trainers
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=["<|begin_of_text|>", "<|end_of_text|>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(CORPUS, trainer)
    bos = tok.token_to_id("<|begin_of_text|>")
    tok.post_processor = processors.TemplateProcessing(single="<|begin_of_text|> $A", pair="<|begin_of_text|> $A <|begin_of_text|> $B",
                                                       special_tokens=[("<|begin_of_text|>", bos)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<|begin_of_text|>", eos_token="<|end_of_text|>")
    os.makedirs(path, exist_ok=True)
    fast.save_pretrained(path)
    return path


INSTRUCTIONS = ["Definition: describe [EXT] then <|protein|> and [EXT] finally [ANSWER]",
                "Short one <|protein|> <|struct|> [PROT] [ANSWER] [EXT]",
                "No description here <|protein|> [ANSWER]"]
TEXTS = [["alpha beta gamma delta " * 30, "Drug: <|drug|> tail words " + "filler " * 10], ["only one description " * 5], []]
