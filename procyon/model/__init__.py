"""`procyon.model`: the engine-backed model classes under the reference's module paths."""
