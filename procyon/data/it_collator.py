"""`procyon.data.it_collator.construct_task_id` (reference: procyon/data/it_collator.py:886-897)."""


def construct_task_id(aaseq_type, text_type, relation_type, task_type):
    """File stem of a task template: `domain_` is spelled out, proteins and peptides share the plain name."""
    kind = aaseq_type.lower()
    if kind == "domain":
        return f"domain_{text_type}_{relation_type}_{task_type}"
    if kind in ("protein", "peptide"):
        return f"{text_type}_{relation_type}_{task_type}"
    raise NotImplementedError("No dataset found for aaseq_type = {}".format(aaseq_type))
