"""Drop-in for `python purejaxql/pqn_gymnax.py +alg=pqn_cartpole` (MLP Q-network, gymnax classic control)."""
import sys

from .run import main

if __name__ == "__main__":
    main(sys.argv[1:], "pqn_cartpole")
