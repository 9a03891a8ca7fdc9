"""Drop-in for `python purejaxql/pqn_craftax.py +alg=pqn_craftax alg.ENV_NAME=Craftax-Classic-Symbolic-v1`
(BatchRenorm MLP Q-network, wrapper-batched env with optimistic resets, 1-step-target loss; pqn_craftax.py:82-468)."""
import sys

from .run import main

if __name__ == "__main__":
    main(sys.argv[1:], "pqn_craftax", script="craftax")
