"""``python -m tamago_amd.selfplay`` - see tamago_amd/selfplay/main.py."""
from tamago_amd.selfplay.main import main

main()
