"""Console + file logger with the reference's line format (reference utility/logging.py:4-14)."""
import os
from datetime import datetime


class Logger():
    def __init__(self, filename, is_debug, path='./logs/'):
        self.filename, self.path, self.log_ = filename, path, not is_debug

    def logging(self, s):
        s = str(s)
        now = datetime.now()
        print(now.strftime('%Y-%m-%d %H:%M: '), s)
        if self.log_:
            os.makedirs(self.path, exist_ok=True)        # the reference crashes if ./logs/ is absent
            with open(os.path.join(self.path, self.filename), 'a+') as f_log:
                f_log.write(now.strftime('%Y-%m-%d %H:%M:  ') + s + '\n')
