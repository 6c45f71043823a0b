"""Visit / time budget (mirror of mcts/time_manager.py:12-163)."""
import time
from enum import Enum

from tamago_amd.board.stone import color_value
from tamago_amd.mcts.constant import CONST_VISITS, CONST_TIME, REMAINING_TIME, VISITS_PER_SEC


class TimeControl(Enum):
    CONSTANT_PLAYOUT = 0
    CONSTANT_TIME = 1
    TIME_CONTROL = 2
    STRICT_PLAYOUT = 3


class TimeManager:
    def __init__(self, mode: TimeControl, constant_visits: int = CONST_VISITS,
                 constant_time: float = CONST_TIME, remaining_time: float = REMAINING_TIME):
        self.mode = mode
        self.constant_visits = constant_visits
        self.constant_time = constant_time
        self.default_time = remaining_time
        self.search_speed = VISITS_PER_SEC
        self.remaining_time = [remaining_time] * 2
        self.time_limit = 0
        self.start_time = 0

    def initialize(self):
        self.remaining_time = [self.default_time] * 2

    def set_search_speed(self, visits: int, consumption_time: float):
        self.search_speed = visits / consumption_time if visits > 0 else VISITS_PER_SEC

    def get_num_visits_threshold(self, color) -> int:
        """Visit budget of the coming search; also arms `time_limit` (time_manager.py:61-83)."""
        fixed_visits = self.mode in (TimeControl.CONSTANT_PLAYOUT, TimeControl.STRICT_PLAYOUT)
        if fixed_visits:
            self.time_limit = 10000.0
            return int(self.constant_visits)
        if self.mode == TimeControl.CONSTANT_TIME:
            self.time_limit = self.constant_time
        else:                                                   # a tenth of the player's clock
            self.time_limit = self.remaining_time[color_value(color) - 1] / 10.0
        return max(int(self.search_speed * self.time_limit), 1)

    def set_remaining_time(self, color, remaining_time: float):
        self.remaining_time[color_value(color) - 1] = remaining_time

    def substract_consumption_time(self, color, consumption_time: float):
        self.remaining_time[color_value(color) - 1] -= consumption_time

    def set_mode(self, mode: TimeControl):
        self.mode = mode

    def start_timer(self):
        self.start_time = time.time()

    def calculate_consumption_time(self) -> float:
        return time.time() - self.start_time

    def is_time_over(self) -> bool:
        return time.time() - self.start_time > self.time_limit

    def is_move_decided(self, root, threshold: int) -> bool:
        """time_manager.py:146-163.  The reference evaluates this after every descent, but
        its inputs only change when a mini-batch is backed up, so the GPU search evaluates
        it once per mini-batch with the same outcome."""
        ordered = sorted(int(v) for v in root.children_visits)
        remaining = threshold - int(root.node_visits)
        cutoff = ordered[-1] - ordered[-2]
        if self.mode == TimeControl.STRICT_PLAYOUT:
            cutoff = 0
        return remaining < cutoff
