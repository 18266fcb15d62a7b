"""A miniature, deterministic ProCyon-Instruct directory ($DATA_DIR) + task templates ($HOME_DIR) for the host-side input builders
(`procyon/data/inference_utils.py`).  Shared by the golden generator (tests/golden/make_golden.py g13, which runs the REFERENCE's
builders over it) and by the CPU tests (which run the shim's builders over an identical tree), so both sides read the same
bytes.  `tables` = the description-column tables of the training run for the datasets of the tree (fixture input data)."""
import json
import os

import pandas as pd

ROWS = {"uniprot": 20000, "omim": 5000, "disgenet": 240000, "pfam": 30000}


def build_tree(root, tasks, tables=None, with_subsets_json=True):
    """-> (data_dir, home_dir).  tasks: {task_id: template dict}; tables: {"ENTITY_DESCRIPTION_NAMES": {...}, "QA_SUBSETS": {...}, ...}
    or None (then two generic description columns per dataset, the layout the older shim tests use)."""
    home, data = os.path.join(root, "home"), os.path.join(root, "data")
    tdir = os.path.join(home, "procyon", "data", "instruct_tune", "tasks")
    os.makedirs(tdir, exist_ok=True)
    for name, t in tasks.items():
        with open(os.path.join(tdir, f"{name}.json"), "w") as f:
            json.dump(t, f)
    prot = os.path.join(data, "integrated_data", "v1", "protein")
    os.makedirs(prot, exist_ok=True)
    n = 20000
    pd.DataFrame({"index": range(n), "protein_id": [f"P{i:05d}" for i in range(n)], "name": [f"GENE{i}" for i in range(n)]}).to_pickle(
        os.path.join(prot, "protein_info_filtered.pkl"))
    # deliberately stored out of order: the reference sorts by "index" before use
    fd = pd.DataFrame({"index": range(n), "function": [f"function of protein {i}" for i in range(n)]}).iloc[::-1]
    fd.to_pickle(os.path.join(prot, "uniprot_functional_descriptions.pkl"))
    for ds, rows in ROWS.items():
        d = os.path.join(data, "integrated_data", "v1", ds)
        os.makedirs(d, exist_ok=True)
        if tables is None:
            cols = ["description_a", "description_b"]
        else:
            cols = []
            for key in ("ENTITY_DESCRIPTION_NAMES", "QA_SUBSETS", "RETRIEVAL_SUBSETS", "CAPTION_SUBSETS"):
                t = tables.get(key, {})
                groups = [t] if key == "ENTITY_DESCRIPTION_NAMES" else list(t.values())
                for g in groups:
                    for c in g.get(ds, []):
                        if c is not None and c not in cols:
                            cols.append(c)
        frame = {"index": range(rows)}
        for ci, c in enumerate(cols):
            # column ci is missing (None) on rows where (row + ci) % 3 == 0: "first non-missing column" differs from row to row
            frame[c] = [None if (i + ci) % 3 == 0 and len(cols) > 1 else f"{ds} {c} {i}" for i in range(rows)]
        frame["other"] = 0
        pd.DataFrame(frame).to_pickle(os.path.join(d, f"{ds}_info_filtered_composed.pkl"))
    if tables is not None and with_subsets_json:
        with open(os.path.join(data, "procyon_column_subsets.json"), "w") as f:
            json.dump(tables, f)
    return data, home


def jsonable(x):
    """model-input dictionaries -> plain JSON (tensors as lists) for comparison"""
    import torch
    if isinstance(x, dict):
        return {k: jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, torch.Tensor):
        return {"__tensor__": x.tolist(), "dtype": str(x.dtype)}
    return x


BUILDER_CASES = [
    # (function, kwargs) -- data_args is supplied by the caller
    ("create_caption_input_simple", dict(input_aaseq_ids=[530], instruction_source_dataset="uniprot", icl_example_number=1)),
    ("create_caption_input_simple", dict(input_aaseq_ids=[530, 7], instruction_source_dataset="UniProt", icl_example_number=2)),
    ("create_caption_input_simple", dict(input_aaseq_ids=[11], instruction_source_dataset="uniprot", icl_example_number=0)),
    ("create_caption_input_simple", dict(input_aaseq_ids=[11], instruction_source_dataset="uniprot", icl_example_number=1,
                                         task_definition="Describe the function of this protein in one sentence.")),
    ("create_caption_input_simple", dict(input_aaseq_ids=[11], instruction_source_dataset="uniprot", icl_example_number=1,
                                         disease_context_augmentation=True)),
    ("create_caption_input_simple", dict(input_aaseq_ids=[11], instruction_source_dataset="uniprot", icl_example_number=1,
                                         input_description="ignored for captions?", task_type="qa")),
    ("create_qa_input_simple", dict(input_aaseq_ids=[77], input_description="query text", instruction_source_dataset="omim")),
    ("create_qa_input_simple", dict(input_aaseq_ids=[77], input_description="query text", instruction_source_dataset="omim", icl_example_number=2)),
    ("create_qa_input_simple", dict(input_aaseq_ids=[77], input_description="query text", instruction_source_dataset="omim", icl_example_number=0)),
    ("create_qa_input_simple", dict(input_aaseq_ids=[77], input_description="q", instruction_source_dataset="omim",
                                    task_definition="Decide whether the protein is linked to the disease.")),
    ("create_qa_input_simple", dict(input_aaseq_ids=[77], input_description="q", instruction_source_dataset="omim",
                                    disease_context_augmentation=True)),
    ("create_qa_input_simple", dict(input_aaseq_ids=[3], input_description="a domain question", instruction_source_dataset="pfam",
                                    aaseq_type="domain")),
    ("create_input_retrieval", dict(input_description="a disease description", instruction_source_dataset="disgenet")),
    ("create_input_retrieval", dict(input_description="a disease description", instruction_source_dataset="DisGeNET", icl_example_number=2)),
    ("create_input_retrieval", dict(input_description="a disease description", instruction_source_dataset="disgenet", icl_example_number=0)),
    ("create_input_retrieval", dict(input_description="d", instruction_source_dataset="disgenet",
                                    task_definition="Find proteins associated with the described condition.")),
]
