#!/usr/bin/env python3
"""Entry point with the reference's file name (selfplay_main.py): one worker shard per GPU.
Same as ``python -m tamago_amd.selfplay``; see tamago_amd/selfplay/main.py for the options."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tamago_amd.selfplay.main import main  # noqa: E402

if __name__ == "__main__":
    main()
