"""``save_checkpoint(args, iteration, model)`` as the reference's drivers call it (src/checkpointing.py:17-22): the engine writes
``<save_dir>/latest_model/mp_rank_00_model_states.pt`` with ``client_state = {"args", "iteration"}``."""


def save_checkpoint(args, iteration, model):
    model.save_checkpoint(args.save_dir, client_state={"args": args, "iteration": iteration}, tag="latest_model")
