"""Helpers the generation callers import (mirrors the reference's src/utils package for this path)."""
