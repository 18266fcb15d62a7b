"""Instruction templates (`procyon/data/instruct_tune/instruct_constructor.py:5-365` of the reference).

One table-driven builder instead of the reference's per-category string literals; `tests/golden/g11_prompts.json` holds the
prompts the reference's own `get_prompt` / `get_prompt_open_def` produce for its task files (all categories, protein / domain /
peptide, PPI and non-PPI, 0-2 in-context examples) and `tests/test_shim_cpu.py` compares character for character.

Placeholders the callers resolve later: `<|protein|>` (soft token), `[PROT]` (retrieval read-out), `[ANSWER]`, `[EXT]` (text
slot), `[CONTEXT]` (optional context slot), `{answer}` (QA target) and, for the open-definition form, `{definition}`.
"""
from __future__ import annotations

_NAMES = {"protein": "Protein", "domain": "Domain", "peptide": "Peptide"}


def aaseq_type_to_prompt(aaseq_type):
    """Display name of an amino-acid-sequence type (instruct_constructor.py:5-16)."""
    key = aaseq_type.lower() if isinstance(aaseq_type, str) else aaseq_type
    return _NAMES.get(key, "Amino acid sequence")


def _definition(task, is_special_definition):
    d = task["Definition"]
    if is_special_definition:
        return d
    for key in ("Relationship Summary", "Biological Summary", "Task-Specific Relationship"):
        d = d.replace("{" + key + "}", task[key])
    return d


def _slots(category, is_ppi, aaseq, query):
    """Body of one example (query=False) or of the instance to complete (query=True), without header / answer."""
    if category == "qa":
        if is_ppi:
            return f"{aaseq} 1: <|protein|>\n{aaseq} 2: <|protein|>\nOutput: [ANSWER] "
        return f"Description: [EXT]\n{aaseq}: <|protein|>\n[CONTEXT]Output: [ANSWER] "
    if category == "retrieval":
        if is_ppi:
            return f"{aaseq} 1: <|protein|> \n{aaseq} 2: [PROT]" if query else f"{aaseq} 1: <|protein|>\n{aaseq} 2: <|protein|>"
        tail = "[PROT]" if query else "<|protein|>"
        return f"[CONTEXT]Description: [EXT]\n{aaseq}: {tail}"
    if category == "caption":
        return f"[CONTEXT]{aaseq}: <|protein|>\nOutput: [ANSWER] [EXT]"
    raise KeyError(category)


def _examples(task, category, polarity, num_examples, is_ppi, sample_examples, aaseq):
    """(text or list of texts, text ids, aaseq ids) of the in-context examples of one polarity."""
    key = "Positive Examples" if polarity == "positive" else "Negative Examples"
    header = "Positive example" if polarity == "positive" else "Negative example"
    ex = task[key]
    n = len(ex) if num_examples is None else num_examples
    ex = ex[:n]
    texts = []
    for i, _ in enumerate(ex):
        if category == "retrieval" and is_ppi and sample_examples:    # the sampled PPI retrieval form drops the numbering
            body = f"{aaseq}: <|protein|>\n{aaseq}: <|protein|>"
        else:
            body = _slots(category, is_ppi, aaseq, query=False)
        if category == "qa":
            body += "yes" if polarity == "positive" else "no"
        texts.append(f"{header} {i + 1}:\n{body}")
    if is_ppi and category != "caption":
        text_ids = []
        aaseq_ids = [a for e in ex for a in (e["aaseq_1"], e["aaseq_2"])]
    else:
        text_ids = [e["text"] for e in ex]
        aaseq_ids = [e["aaseq"] for e in ex]
    return (texts if sample_examples else "\n".join(texts)), text_ids, aaseq_ids


def _build(task, definition, num_examples, is_ppi, sample_examples, aaseq_type):
    aaseq = aaseq_type_to_prompt(aaseq_type)
    cat = task["CATEGORY"]
    if cat == "caption":
        assert is_ppi == False, "Cannot use PPI with caption task"   # noqa: E712 (the reference's assertion text)
    pos, pos_t, pos_a = _examples(task, cat, "positive", num_examples, is_ppi, sample_examples, aaseq)
    neg, neg_t, neg_a = (None, [], [])
    if cat == "qa":
        neg, neg_t, neg_a = _examples(task, cat, "negative", num_examples, is_ppi, sample_examples, aaseq)
    query = "Now, complete the following instance:\n" + _slots(cat, is_ppi, aaseq, query=True)
    if cat == "qa":
        query += "{answer}"
    if sample_examples:   # examples are spliced in later by the caller: placeholders instead of text
        holders = "{positive_examples}{negative_examples}\n" if cat == "qa" else "{positive_examples}\n"
        lead = "\n" if cat == "caption" else ""
        prompt = f"Definition: {definition}" + holders + lead + query
    else:
        parts = [f"Definition: {definition}", pos] + ([neg] if cat == "qa" else []) + [query]
        prompt = "\n".join(parts)
    text_ids = [] if (is_ppi and cat != "caption") else pos_t + neg_t
    return prompt, pos, neg, text_ids, pos_a + neg_a


def get_prompt(task, num_examples=None, is_special_definition=False, is_ppi=False, sample_examples=False, aaseq_type=None):
    """-> (prompt, positive example text, negative example text | None, example text ids, example aaseq ids)
    (instruct_constructor.py:111-235)."""
    return _build(task, _definition(task, is_special_definition), num_examples, is_ppi, sample_examples, aaseq_type)


def get_prompt_open_def(task, num_examples=None, is_special_definition=False, is_ppi=False, sample_examples=False, aaseq_type=None):
    """The same prompt with `{definition}` left open, plus the task's own definition
    -> (prompt, true definition, positive, negative | None, text ids, aaseq ids) (instruct_constructor.py:237-365)."""
    assert not sample_examples, "Not supported"
    prompt, pos, neg, text_ids, aaseq_ids = _build(task, "{definition}", num_examples, is_ppi, sample_examples, aaseq_type)
    return prompt, _definition(task, is_special_definition), pos, neg, text_ids, aaseq_ids
