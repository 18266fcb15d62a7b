"""`procyon.inference.settings.logger` (reference: procyon/inference/settings.py): a loguru logger writing to stdout at
$PROCYON_LOG_LEVEL; where loguru is not installed, a stdlib logger with the same call surface (.debug/.info/.warning/.error)."""
import os
import sys

LOG_LEVEL = os.getenv("PROCYON_LOG_LEVEL")


def construct_logger(log_level="INFO"):
    try:
        from loguru import logger
        logger.remove()
        logger.add(sys.stdout, level=log_level)
        return logger
    except ImportError:
        import logging
        lg = logging.getLogger("procyon")
        if not lg.handlers:
            h = logging.StreamHandler(sys.stdout)
            h.setFormatter(logging.Formatter("%(asctime)s | %(levelname)-8s | %(message)s"))
            lg.addHandler(h)
        lg.setLevel(getattr(logging, str(log_level).upper(), logging.INFO))
        return lg


logger = construct_logger(LOG_LEVEL or "INFO")
logger.info(f"Log level: {LOG_LEVEL}")
