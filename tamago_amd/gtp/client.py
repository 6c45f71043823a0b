"""Go Text Protocol front end on the device-resident search (next row 8(f).2; mirrors the
command set and the response texts of gtp/client.py:31-600 that reach the hot path):
genmove / lz-genmove_analyze / cgos-genmove_analyze -> MCTSTree.search_best_move (or the
Gumbel search), lz-analyze / cgos-analyze -> MCTSTree.ponder, plus the board bookkeeping
commands a GTP controller needs around them (incl. fixed_handicap).  Not carried over: gogui colour
maps, tree dump, animation.  tests/test_gpu_gtp.py replays a session recorded from the reference's
command loop byte for byte."""
import sys
from typing import Callable, Dict, List

from tamago_amd.board.constant import PASS, RESIGN
from tamago_amd.board.coordinate import Coordinate
from tamago_amd.board.go_board import GoBoard
from tamago_amd.board.stone import Stone
from tamago_amd.mcts.time_manager import TimeControl, TimeManager
from tamago_amd.mcts.tree import MCTSTree
from tamago_amd.sgf.reader import SGFReader

PROGRAM_NAME = "TamaGo"            # program.py:3-4
VERSION = "0.10.0"
PROTOCOL_VERSION = "2"


class GtpClient:
    def __init__(self, board_size: int, superko: bool, network, komi: float = 7.0,
                 mode: TimeControl = TimeControl.CONSTANT_PLAYOUT, visits: int = 1000,
                 const_time: float = 5.0, time: float = 0.0, batch_size: int = 256,
                 tree_size: int = 65536, cgos_mode: bool = False, use_sequential_halving: bool = False,
                 stdin=None, stdout=None):
        """`network`: a DualNet (device forward) or any object with the DualNet host API."""
        self.superko = superko
        self.komi = komi
        self.board = GoBoard(board_size=board_size, komi=komi, check_superko=superko)
        self.coordinate = Coordinate(board_size=board_size)
        self.history: List = []                       # (pos, colour) since the last clear_board
        self.use_sequential_halving = use_sequential_halving
        if mode in (TimeControl.CONSTANT_PLAYOUT, TimeControl.STRICT_PLAYOUT):
            self.time_manager = TimeManager(mode=mode, constant_visits=visits)
        elif mode is TimeControl.CONSTANT_TIME:
            self.time_manager = TimeManager(mode=mode, constant_time=const_time)
        else:
            self.time_manager = TimeManager(mode=mode, remaining_time=time)
        self.mcts = MCTSTree(network=network, batch_size=batch_size, tree_size=tree_size, cgos_mode=cgos_mode)
        self.stdin = stdin
        self.stdout = stdout
        self.command_id = ""
        self.commands: Dict[str, Callable[[List[str]], None]] = {
            "version": lambda a: self._ok(VERSION),
            "protocol_version": lambda a: self._ok(PROTOCOL_VERSION),
            "name": lambda a: self._ok(PROGRAM_NAME),
            # (gtp/client.py:116-125: an unknown command is a FAILURE "unknown command", not "= false")
            "known_command": lambda a: self._ok("true") if a and a[0] in self.commands else self._fail("unknown command"),
            "list_commands": lambda a: self._ok("\n".join(self.commands)),
            "komi": self._komi,
            "get_komi": lambda a: self._ok(str(self.board.get_komi())),
            "play": self._play,
            "undo": self._undo,
            "genmove": self._genmove,
            "boardsize": self._boardsize,
            "clear_board": self._clear_board,
            "time_settings": self._time_settings,
            "time_left": self._time_left,
            "fixed_handicap": self._fixed_handicap,
            "showboard": self._showboard,
            "loadsgf": self._loadsgf,
            "final_score": lambda a: self._ok("?"),
            "lz-analyze": lambda a: self._analyze("lz", a),
            "cgos-analyze": lambda a: self._analyze("cgos", a),
            "lz-genmove_analyze": lambda a: self._genmove_analyze("lz", a),
            "cgos-genmove_analyze": lambda a: self._genmove_analyze("cgos", a),
            "quit": None,
        }

    # ---- responses (gtp/client.py:601-617) ------------------------------------------------
    def _write(self, text: str):
        out = self.stdout or sys.stdout
        out.write(text)
        out.flush()

    def _ok(self, response: str, ongoing: bool = False):
        self._write(f"={self.command_id} " + response + ("\n" if ongoing else "\n\n"))

    def _fail(self, response: str):
        self._write(f"?{self.command_id} " + response + "\n\n")

    @staticmethod
    def _color(text: str):
        first = text.lower()[:1]
        return Stone.BLACK if first == "b" else (Stone.WHITE if first == "w" else None)

    # ---- board bookkeeping -----------------------------------------------------------------
    def _komi(self, args):
        try:
            self.komi = float(args[0])
        except (IndexError, ValueError):
            return self._fail("komi float")
        self.board.set_komi(self.komi)
        self._ok("")

    def _put(self, pos: int, color):
        self.board.put_stone(pos, color)
        self.history.append((pos, color))

    def _play(self, args):
        color = self._color(args[0]) if len(args) >= 2 else None
        if color is None:
            return self._fail("play color pos")
        pos = self.coordinate.convert_from_gtp_format(args[1])
        if pos != PASS and not self.board.is_legal(pos, color):
            self._write(f"illigal {args[0]} {args[1]}\n")         # (sic, gtp/client.py:167)
        self._put(pos, color)
        self._ok("")

    def _rebuild(self, history, handicaps=()):
        self.board = GoBoard(board_size=self.board.get_board_size(), komi=self.komi, check_superko=self.superko)
        self.history = []
        for pos in handicaps:                         # go_board.py:554-560: handicap stones first, then the moves
            self.board.put_handicap_stone(pos, Stone.BLACK)
        for pos, color in history:
            self._put(pos, color)

    def _fixed_handicap(self, args):
        """gtp/client.py:342-366."""
        if self.board.moves > 1 or len(self.board.get_handicap_history()) > 1:
            return self._fail("board not empty")
        from tamago_amd.board.handicap import get_handicap_coordinates
        size = self.board.get_board_size()
        points = get_handicap_coordinates(size, int(args[0]))
        if points is None:
            return self._fail(f"size {size}, handicaps {args[0]} is not supported")
        for point in points:
            self.board.put_handicap_stone(self.coordinate.convert_from_gtp_format(point), Stone.BLACK)
        self._ok(" ".join(points))

    def _undo(self, args):
        if not self.history:
            return self._fail("cannot undo")
        self._rebuild(self.history[:-1], self.board.get_handicap_history())
        self._ok("")

    def _boardsize(self, args):
        try:
            size = int(args[0])
        except (IndexError, ValueError):
            return self._fail("boardsize int")
        net_size = getattr(self.mcts.network, "board_size", None)
        if size not in (9, 13, 19) or (net_size is not None and net_size != size):
            # the reference answers an out-of-range size with "?"; here the size must also be the
            # one the resident network was built for (a [B,82] policy cannot serve a 19x19 tree)
            return self._fail("unacceptable size")
        self.board = GoBoard(board_size=size, komi=self.komi, check_superko=self.superko)
        self.coordinate = Coordinate(board_size=size)
        self.history = []
        self.time_manager.initialize()
        self._ok("")

    def _clear_board(self, args):
        self._rebuild([])
        self.time_manager.initialize()
        self._ok("")

    def _time_settings(self, args):
        # gtp/client.py:255-265: the main time only; the time-control MODE stays what the client was started with
        try:
            for color in (Stone.BLACK, Stone.WHITE):
                self.time_manager.set_remaining_time(color, float(args[0]))
        except (IndexError, ValueError):
            return self._fail("time_settings main_time byo_yomi_time byo_yomi_stones")
        self._ok("")

    def _time_left(self, args):
        color = self._color(args[0]) if len(args) >= 2 else None
        if color is None:
            return self._fail("time_left color time stones")
        self.time_manager.set_remaining_time(color, float(args[1]))
        self._ok("")

    def _showboard(self, args):
        size = self.board.get_board_size()
        data = self.board.get_board_data()
        rows = ["".join(".XO"[v] for v in data[y * size:(y + 1) * size]) for y in range(size)]
        sys.stderr.write("\n".join(f"{size - y:2d} {row}" for y, row in enumerate(rows)) + "\n")
        self._ok("")

    def _loadsgf(self, args):
        if not args:
            return self._fail("loadsgf filename [move_number]")
        try:
            sgf = SGFReader(args[0], self.board.get_board_size())
            upto = int(args[1]) - 1 if len(args) > 1 else 9999
        except (OSError, ValueError):
            return self._fail(f"cannot load {args[0]}")
        self.komi = sgf.komi
        self._rebuild([])
        self.board.set_komi(self.komi)
        for i in range(min(upto, sgf.get_n_moves())):
            self._put(sgf.get_move_data(i), sgf.get_color(i))
        self._ok("")

    # ---- the hot path ---------------------------------------------------------------------
    def _search(self, color, analysis_query):
        if self.use_sequential_halving and not analysis_query:
            return self.mcts.generate_move_with_sequential_halving(self.board, color, self.time_manager, False)
        return self.mcts.search_best_move(self.board, color, self.time_manager, analysis_query)

    def _genmove(self, args):
        color = self._color(args[0]) if args else None
        if color is None:
            return self._fail("genmove color")
        pos = self._search(color, {})
        if pos != RESIGN:
            self._put(pos, color)
        self._ok(self.coordinate.convert_to_gtp_format(pos))

    def _analyze_args(self, args):
        """[color] [interval] [centiseconds] (gtp/client.py:368-405); interval < 0 = bad arguments."""
        args = list(args)
        to_move = self.board.get_to_move()
        interval = 0.0
        if args and self._color(args[0]) is not None and not args[0].isdigit() and args[0] != "interval":
            to_move = self._color(args.pop(0))
        if args and args[0] == "interval":
            if len(args) == 1:
                return to_move, -1.0
            args.pop(0)
        if args and args[0].isdigit():
            interval = int(args.pop(0)) / 100
        return (to_move, -1.0) if args else (to_move, interval)

    def _analyze(self, mode, args):
        to_move, interval = self._analyze_args(args)
        if interval < 0:
            return self._fail(f"{mode}-analyze [color] [interval]")
        self._ok("", ongoing=True)
        self.mcts.ponder(self.board, to_move, {"mode": mode, "interval": interval, "ponder": True})
        self._write("\n")

    def _genmove_analyze(self, mode, args):
        color, interval = self._analyze_args(args)
        if interval < 0:
            return self._fail(f"{mode}-analyze [color] [interval]")
        self._ok("", ongoing=True)
        pos = self._search(color, {"mode": mode, "interval": interval, "ponder": False})
        if pos != RESIGN:
            self._put(pos, color)
        self._write(f"play {self.coordinate.convert_to_gtp_format(pos)}\n\n")

    # ---- main loop (gtp/client.py:487-600) -------------------------------------------------
    def run(self):
        source = self.stdin or sys.stdin
        while True:
            line = source.readline()
            if not line:
                break
            words = line.rstrip().split(" ")
            self.command_id = ""
            if words and words[0].isdigit():                      # optional command id (GTP 2, 2.5)
                self.command_id = words.pop(0)
            if not words or not words[0]:
                continue
            name, args = words[0], words[1:]
            if name == "quit":
                self._ok("")
                break
            handler = self.commands.get(name)
            if handler is None:
                self._fail("unknown_command")
            else:
                handler(args)
