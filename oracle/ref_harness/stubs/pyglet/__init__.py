"""Empty stand-in: gym_go/rendering.py:2 imports pyglet at module import; nothing is drawn."""
