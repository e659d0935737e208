"""Names of the six state planes and of the colour / wildcard constants - facts of the `[6, N, N]` state format that
callers of the reference address by name (gym_go/govars.py:1-11), so the names and values are part of the drop-in API."""
# plane order inside a state: stones of either colour, side to move, the next mover's illegal points, "previous player
# passed", "game over"
BLACK, WHITE, TURN_CHNL, INVD_CHNL, PASS_CHNL, DONE_CHNL = range(6)
NUM_CHNLS = DONE_CHNL + 1

# values a `player` argument may take besides BLACK / WHITE
ANYONE, NOONE = None, -1
