"""`procyon` import surface over the MI355X engine (`procyon_amd`).

The reference's entry points (scripts/caption_bulk.py, scripts/protein_retrieval_disease_pheno.py, procyon/app/main.py, the
evaluate framework, the example notebooks) import a fixed set of names from the `procyon` package.  This
package provides exactly those import paths, written from the interface: the model classes come from `procyon_amd.model`
(HIP kernels behind the C ABI), the host-side helpers (prompt construction, input dicts, retrieval ranking, argument shells,
logger) are restated here.  Nothing reads a dataset, a checkpoint or an environment variable at import time; the data files
of ProCyon-Instruct (DATA_DIR) and the task templates (HOME_DIR/procyon/data/instruct_tune/tasks) are opened when a function
that needs them is called.

Put the repository root on PYTHONPATH *instead of* the reference checkout to switch an entry point over.
"""
__all__ = ["model", "inference", "data", "evaluate", "training"]
