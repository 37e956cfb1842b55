"""The reference's adaptive food-count curriculum on the batched simulator.

Restates `ContinuousTrainer._update_adaptive_difficulty` (src/salp/training/continuous_trainer.py:375-415): the food
collection rate of each finished episode (food collected / foods of that episode) enters a window of the last 10
episodes; once the window is full, a mean above 0.6 removes one food from later episodes and a mean below 0.25 adds
one, within [2, 12]; after a change the window is cleared.  The change is the attribute write
`env.base_num_food_items = k` (:409-411), which the C ABI exposes as `salp_vec_set_base_num_food`.

With thousands of envs many episodes end in the same vector step; they enter the window in env order, as they would
have one after the other in the single-env trainer.
"""
from __future__ import annotations

from typing import List, Sequence


class AdaptiveFoodCurriculum:
    def __init__(self, env, min_food_count: int = 2, max_food_count: int = 12, window: int = 10,
                 harder_above: float = 0.6, easier_below: float = 0.25):
        self.env = env
        slots = int(getattr(getattr(env, "cfg", None), "num_food_items", max_food_count))
        if int(max_food_count) > slots:
            # the reference's write simply grows the food list; the batched state has a fixed number of food slots
            raise ValueError(f"the env was created with num_food_items={slots} food slots, the curriculum may ask for up "
                             f"to {max_food_count}: create the env with num_food_items >= max_food_count "
                             f"(and start lower with `env.base_num_food_items = k`)")
        self.min_food_count, self.max_food_count = int(min_food_count), int(max_food_count)
        self.window, self.harder_above, self.easier_below = int(window), float(harder_above), float(easier_below)
        self.current_food_count = int(env.base_num_food_items)
        self.recent_food_collection_rates: List[float] = []
        self.changes: List[int] = []

    def record_episode(self, food_collected: int, total_food_available: int) -> bool:
        """One finished episode (continuous_trainer.py:377-412).  Returns True if the food count changed."""
        rate = food_collected / max(1, total_food_available)
        self.recent_food_collection_rates.append(rate)
        if len(self.recent_food_collection_rates) > self.window:
            self.recent_food_collection_rates.pop(0)
        if len(self.recent_food_collection_rates) < self.window:
            return False
        avg = sum(self.recent_food_collection_rates) / len(self.recent_food_collection_rates)
        old = self.current_food_count
        if avg > self.harder_above:
            self.current_food_count = max(self.min_food_count, self.current_food_count - 1)
        elif avg < self.easier_below:
            self.current_food_count = min(self.max_food_count, self.current_food_count + 1)
        if self.current_food_count == old:
            return False
        self.env.base_num_food_items = self.current_food_count
        self.recent_food_collection_rates = []
        self.changes.append(self.current_food_count)
        return True

    def record_finished(self, food_collected: Sequence[int], total_food_available: Sequence[int]) -> int:
        """All episodes that ended in one vector step (`info["food_collected"]` and `env.num_food_items` of the
        finished envs, taken BEFORE the step's autoreset shows in the state).  Returns the number of changes."""
        n = 0
        for fc, tot in zip(food_collected, total_food_available):
            n += bool(self.record_episode(int(fc), int(tot)))
        return n
