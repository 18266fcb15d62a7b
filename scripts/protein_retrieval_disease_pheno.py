"""Protein retrieval for one disease / phenotype description over the MI355X engine: the command line of the reference's
scripts/protein_retrieval_disease_pheno.py (/root/reference/scripts/protein_retrieval_disease_pheno.py:1-85 -- same arguments, same
environment variables, same log lines, same return value), on `procyon.inference.retrieval_utils.{startup_retrieval, do_retrieval}`."""
import argparse
import os

from procyon.inference.retrieval_utils import do_retrieval, startup_retrieval
from procyon.inference.settings import logger


def single_retrieval(task_desc_infile, disease_desc_infile, instruction_source_dataset, inference_bool=True):
    """One retrieval run: rank every protein of $CHECKPOINT_PATH/protein_target_embeddings.pkl for the disease description.
    inference_bool=False checks the command line without loading the model.  -> data frame (uniprot_id, name, sim_score) or None."""
    if not os.getenv("CHECKPOINT_PATH"):
        raise EnvironmentError("CHECKPOINT_PATH environment variable not set")
    model, device, data_args, all_protein_embeddings = startup_retrieval(inference_bool)
    results_df = do_retrieval(model, data_args, device, instruction_source_dataset, all_protein_embeddings, inference_bool=inference_bool,
                              task_desc_infile=task_desc_infile, disease_desc_infile=disease_desc_infile)
    if results_df is not None:
        logger.info(f"top results: {results_df.head(10).to_dict(orient='records')}")
    logger.info("DONE WITH ALL WORK")
    return results_df


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--task_desc_infile", type=str, help="Description of the task.")
    parser.add_argument("--disease_desc_infile", type=str, help="Description of the task.")
    parser.add_argument("--inference_bool", action="store_false", default=True,
                        help="OPTIONAL; choose this if you do not intend to do inference or load the model. Loading the model "
                             "is time-consuming, so consider using this to test that the CLI works.")
    parser.add_argument("--instruction_source_dataset", type=str, choices=["disgenet", "omim"], default="omim",
                        help="Dataset source for instructions - either 'disgenet' or 'omim'")
    args = parser.parse_args()
    single_retrieval(args.task_desc_infile, args.disease_desc_infile, args.instruction_source_dataset, args.inference_bool)
