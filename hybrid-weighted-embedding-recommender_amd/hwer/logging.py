"""Same contract as the reference's hwer/logging.py:3-13 — a stdlib logger whose level comes from $LOGLEVEL."""
import logging
import os

logging.basicConfig(format="[PID: %(process)d] [%(asctime)s] [%(levelname)s] [%(name)s]: %(message)s",
                    level=os.environ.get("LOGLEVEL", "INFO"), datefmt="%Y-%m-%d %H:%M:%S")


def getLogger(name, level=None):
    log = logging.getLogger(name)
    log.setLevel(os.environ.get("LOGLEVEL", "INFO") if level is None else level)
    return log
