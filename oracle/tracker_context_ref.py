"""Pure-Python restatement of the reference's tracker control layer — TEST INFRASTRUCTURE ONLY.

Follows, line by line:
  AppState        /root/reference/src/app_state.rs:1-6
  SelectionState  /root/reference/src/selection_state.rs:9-46
  TimingStats     /root/reference/src/timing_stats.rs:3-61
  TrackerContext  /root/reference/src/tracker_context.rs:7-167
The reference has no tests for these; the known-answer cases in
tests/test_host_control.py are derived from the source (strict `> 0.25`, reset when the Lost
counter already exceeds 60, bbox sides clamped to >= 20, rings of 120 samples).
"""
from __future__ import annotations

from collections import deque

SELECTING, TRACKING, LOST = 0, 1, 2
MOVE_UP, MOVE_DOWN, MOVE_LEFT, MOVE_RIGHT, CONFIRM, CANCEL, QUIT = range(7)
MOVING_TO_START, SELECTING_AREA = 0, 1


class SelectionState:
    def __init__(self, width, height):                     # selection_state.rs:21-31
        self.cursor_x, self.cursor_y = width // 2, height // 2
        self.start_x, self.start_y = width // 2, height // 2
        self.phase = MOVING_TO_START
        self.step, self.fast_step = 10, 50

    def move_cursor(self, dx, dy, fast, width, height):     # :33-37
        step = self.fast_step if fast else self.step
        self.cursor_x = min(max(self.cursor_x + dx * step, 0), width - 1)
        self.cursor_y = min(max(self.cursor_y + dy * step, 0), height - 1)

    def get_bbox(self):                                     # :39-45
        x = min(self.start_x, self.cursor_x)
        y = min(self.start_y, self.cursor_y)
        w = max(abs(self.start_x - self.cursor_x), 20)
        h = max(abs(self.start_y - self.cursor_y), 20)
        return (x, y, w, h)


class TimingStats:
    def __init__(self):                                     # timing_stats.rs:9-16
        self.intervals, self.conv_times, self.track_times = deque(), deque(), deque()

    @staticmethod
    def _push(q, v):
        if len(q) >= 120:
            q.popleft()
        q.append(v)

    def add_interval(self, v):                              # :18-23
        self._push(self.intervals, v)

    def add_times(self, conv, track):                       # :25-34
        self._push(self.conv_times, conv)
        self._push(self.track_times, track)

    def fps(self):                                          # :36-46
        if not self.intervals:
            return 0.0
        avg = sum(self.intervals) / len(self.intervals)
        return 1_000_000.0 / avg if avg > 0.0 else 0.0

    def avg_conv_ms(self):                                  # :48-53
        return sum(self.conv_times) / len(self.conv_times) / 1000.0 if self.conv_times else 0.0

    def avg_track_ms(self):                                 # :55-60
        return sum(self.track_times) / len(self.track_times) / 1000.0 if self.track_times else 0.0


class TrackerContext:
    """tracker: object with init(frame, bbox) and update(frame) -> (success, score, bbox) or raising
    for the reference's Err arm."""

    def __init__(self, tracker, width, height):             # tracker_context.rs:19-34
        self.tracker = tracker
        self.state, self.lost_frames = SELECTING, 0
        self.selection = SelectionState(width, height)
        self.current_bbox, self.current_score = None, 0.0
        self.frame_width, self.frame_height = width, height
        self.pending_confirm = False

    def handle_command(self, cmd, fast=False):              # :36-61
        if cmd == MOVE_UP:
            self.selection.move_cursor(0, -1, fast, self.frame_width, self.frame_height)
        elif cmd == MOVE_DOWN:
            self.selection.move_cursor(0, 1, fast, self.frame_width, self.frame_height)
        elif cmd == MOVE_LEFT:
            self.selection.move_cursor(-1, 0, fast, self.frame_width, self.frame_height)
        elif cmd == MOVE_RIGHT:
            self.selection.move_cursor(1, 0, fast, self.frame_width, self.frame_height)
        elif cmd == CONFIRM:
            self.pending_confirm = True
        elif cmd == CANCEL:
            self.state, self.lost_frames = SELECTING, 0
            self.selection = SelectionState(self.frame_width, self.frame_height)
            self.current_bbox = None

    def process_frame(self, frame):                         # :64-155
        if self.state == SELECTING:
            if self.pending_confirm:
                self.pending_confirm = False
                if self.selection.phase == MOVING_TO_START:             # :71-80
                    self.selection.start_x = self.selection.cursor_x
                    self.selection.start_y = self.selection.cursor_y
                    self.selection.phase = SELECTING_AREA
                else:                                                   # :81-110
                    bbox = self.selection.get_bbox()
                    self.tracker.init(frame, bbox)                      # :88
                    try:
                        success, score, rb = self.tracker.update(frame)  # :90 same frame
                    except Exception:                                   # :105-109
                        self.selection = SelectionState(self.frame_width, self.frame_height)
                        return None
                    if success and score > 0.25:                        # :93
                        self.current_bbox = tuple(rb)
                        self.current_score = score
                        self.state = TRACKING
                        return self.current_bbox
                    self.selection = SelectionState(self.frame_width, self.frame_height)
            return None
        if self.state == TRACKING:                                      # :115-140
            self.pending_confirm = False
            try:
                success, score, rb = self.tracker.update(frame)
            except Exception:
                self.state, self.lost_frames = LOST, 0                  # :136
                return None
            if success and score > 0.25:                                # :122
                self.current_bbox = tuple(rb)
                self.current_score = score
                return self.current_bbox
            self.state, self.lost_frames = LOST, 0
            self.current_score = 0.0
            return None
        # LOST                                                          # :142-153
        self.pending_confirm = False
        if self.lost_frames > 60:
            self.state, self.lost_frames = SELECTING, 0
            self.selection = SelectionState(self.frame_width, self.frame_height)
            self.current_bbox = None
        else:
            self.lost_frames += 1
        return None

    def state_name(self):                                   # :157-166
        if self.state == SELECTING:
            return "SELECT START" if self.selection.phase == MOVING_TO_START else "SELECT END"
        return "TRACKING" if self.state == TRACKING else "LOST"
