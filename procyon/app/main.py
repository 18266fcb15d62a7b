"""Protein-retrieval HTTP service (reference: procyon/app/main.py:33-80): POST /retrieve {task_desc, disease_desc,
instruction_source_dataset, k} -> {"results": [{uniprot_id, name, sim_score}, ...]} ranked best first.

The model is loaded once at start-up through `startup_retrieval` (engine-backed `UnifiedProCyon` + the cached target matrix of
the checkpoint); each request is one prompt prefill + one ranking launch on the GPU.  Run: `python -m procyon.app.main`."""
import os
from typing import Optional

from fastapi import FastAPI, HTTPException
from pydantic import BaseModel, Field

from procyon.inference.retrieval_utils import do_retrieval, startup_retrieval
from procyon.inference.settings import logger

app = FastAPI()
state = {"model": None, "device": None, "data_args": None, "all_protein_embeddings": None}
REQUIRED_ENV = ("CHECKPOINT_PATH", "HOME_DIR", "DATA_DIR", "LLAMA3_PATH")


class RetrievalRequest(BaseModel):
    task_desc: str = Field(description="The task description.")
    disease_desc: str = Field(description="The disease description.")
    instruction_source_dataset: str = Field(description="Dataset source for instructions - either 'disgenet' or 'omim'")
    k: Optional[int] = Field(default=None, description="Number of top results to return. If None, returns all results", ge=1)


@app.on_event("startup")
async def startup_event():
    for var in REQUIRED_ENV:
        if not os.getenv(var):
            raise EnvironmentError(f"{var} environment variable not set")
    state["model"], state["device"], state["data_args"], state["all_protein_embeddings"] = startup_retrieval(inference_bool=True)
    logger.info("Model loaded and ready")


@app.post("/retrieve")
async def retrieve_proteins(request: RetrievalRequest):
    if any(state[k] is None for k in ("model", "device", "data_args", "all_protein_embeddings")):
        raise HTTPException(status_code=500, detail="Model not initialized")
    try:
        df = do_retrieval(model=state["model"], data_args=state["data_args"], device=state["device"],
                          instruction_source_dataset=request.instruction_source_dataset,
                          all_protein_embeddings=state["all_protein_embeddings"], task_desc=request.task_desc,
                          disease_desc=request.disease_desc)
    except ValueError as e:
        raise HTTPException(status_code=422, detail=str(e))
    df = df.fillna("")
    rows = df if request.k is None else df.head(request.k)
    return {"results": rows.to_dict(orient="records")}


if __name__ == "__main__":
    import uvicorn
    uvicorn.run(app, host="0.0.0.0", port=8000)
