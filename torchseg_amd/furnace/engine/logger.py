"""Root-logger factory with the surface of furnace/engine/logger.py:82-99:
`get_logger(log_dir=None, log_file=None, formatter=LogFormatter)`; level from
ENGINE_LOGGING_LEVEL (logger.py:14-15).  Unlike the reference this module does
not import utils.pyt_utils, which removes its circular import (logger.py:11 <->
pyt_utils.py:12)."""
import logging
import os
import sys

_LEVEL = getattr(logging, os.getenv('ENGINE_LOGGING_LEVEL', 'INFO').upper(), logging.INFO)

_COLORS = {logging.DEBUG: '\x1b[32m', logging.WARNING: '\x1b[1;31m', logging.ERROR: '\x1b[1;4;31m'}


class LogFormatter(logging.Formatter):
    """date + level prefix, coloured on a tty, plain in files (logger.py:18-79)."""
    log_fout = None
    date_full = '[%(asctime)s %(lineno)d@%(filename)s:%(name)s] '
    date = '%(asctime)s '

    def format(self, record):
        prefix = {logging.DEBUG: 'DBG', logging.WARNING: 'WRN', logging.ERROR: 'ERR'}.get(record.levelno, '')
        msg = super().format(record)
        if prefix:
            color = _COLORS.get(record.levelno, '') if sys.stdout.isatty() else ''
            reset = '\x1b[0m' if color else ''
            return '{}{} {}{}'.format(color, prefix, msg, reset)
        return msg


def get_logger(log_dir=None, log_file=None, formatter=LogFormatter):
    logger = logging.getLogger()
    logger.setLevel(_LEVEL)
    for h in list(logger.handlers):
        if getattr(h, '_tsg', False):
            logger.removeHandler(h)
    fmt = formatter(fmt='%(asctime)s %(message)s', datefmt='%d %H:%M:%S')
    if log_dir and log_file:
        os.makedirs(log_dir, exist_ok=True)
        fh = logging.FileHandler(log_file, mode='a')
        fh.setLevel(logging.INFO)
        fh.setFormatter(fmt)
        fh._tsg = True
        logger.addHandler(fh)
    sh = logging.StreamHandler()
    sh.setFormatter(fmt)
    sh.setLevel(0)
    sh._tsg = True
    logger.addHandler(sh)
    return logger
