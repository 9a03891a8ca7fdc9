"""Drop-in for `python purejaxql/pqn_minatar.py +alg=pqn_minatar` (CNN Q-network, MinAtar)."""
import sys

from .run import main

if __name__ == "__main__":
    main(sys.argv[1:], "pqn_minatar")
