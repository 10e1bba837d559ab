"""stdout + file logger (reference: io/logger.py); ANSI colour codes are stripped for the file."""
import logging
import os
import re
import sys
from typing import Optional

__all__ = ["Logger"]

_ANSI = re.compile(r"\x1B(?:[@-Z\\-_]|\[[0-?]*[ -/]*[@-~])")


class _StripColour(logging.Filter):
    def filter(self, record):
        record.msg = _ANSI.sub("", str(record.msg))
        return True


class Logger(object):
    def __init__(self, filename: Optional[str] = None):
        self._logger_name = "scikit-recommender-logger" if filename is None else filename
        self.logger = logging.getLogger(self._logger_name)
        self.logger.setLevel(logging.DEBUG)
        self.logger.propagate = False
        fmt = logging.Formatter("%(asctime)s.%(msecs)03d: %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
        console = logging.StreamHandler(sys.stdout)
        console.setFormatter(fmt)
        self.logger.addHandler(console)
        if filename is not None:
            folder = os.path.dirname(filename)
            if folder:
                os.makedirs(folder, exist_ok=True)
            fh = logging.FileHandler(filename)
            fh.setFormatter(fmt)
            fh.addFilter(_StripColour())
            self.logger.addHandler(fh)

    @property
    def logger_name(self) -> str:
        return self._logger_name

    def _emit(self, level, message):
        self.logger.log(level, message)
        for h in self.logger.handlers:
            h.flush()

    def debug(self, message):
        self._emit(logging.DEBUG, message)

    def info(self, message):
        self._emit(logging.INFO, message)

    def warning(self, message):
        self._emit(logging.WARNING, message)

    def error(self, message):
        self._emit(logging.ERROR, message)

    def critical(self, message):
        self._emit(logging.CRITICAL, message)
