// tracker_context.hpp — REPLAY HARNESS (not product logic; libvittrack_hip.so does not link it).
// C++ mirror of the reference's tracker control layer:
//   AppState        /root/reference/src/app_state.rs:1-6
//   UserCommand     /root/reference/src/user_commands.rs:1-10
//   SelectionState  /root/reference/src/selection_state.rs:1-46
//   TimingStats     /root/reference/src/timing_stats.rs:1-61
//   TrackerContext  /root/reference/src/tracker_context.rs:1-167
// Same names, same argument meaning, same thresholds (score > 0.25 strictly; auto-reset when a
// Lost counter already exceeds 60), same init-then-update on one frame.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <memory>
#include <optional>
#include <string>

#include "vit_tracker.hpp"

namespace host {

using vit_tracker::BBox;
using vit_tracker::FrameView;
using vit_tracker::ITracker;
using vit_tracker::TrackResult;

// the reference prints progress with println!("...\r"); here only when VT_HOST_VERBOSE is set
inline void log_line(const std::string& s) {
    static const bool on = getenv("VT_HOST_VERBOSE") != nullptr;
    if (on) { fputs(s.c_str(), stdout); fputs("\r\n", stdout); }
}

// src/app_state.rs:1-6
struct AppState {
    enum Kind { Selecting = 0, Tracking = 1, Lost = 2 };
    Kind kind = Selecting;
    uint64_t frames = 0;  // Lost { frames }
    static AppState selecting() { return AppState{Selecting, 0}; }
    static AppState tracking() { return AppState{Tracking, 0}; }
    static AppState lost(uint64_t f) { return AppState{Lost, f}; }
};

// src/user_commands.rs:1-10
struct UserCommand {
    enum Kind { MoveUp = 0, MoveDown = 1, MoveLeft = 2, MoveRight = 3, Confirm = 4, Cancel = 5, Quit = 6 };
    Kind kind = Quit;
    bool fast = false;
};

// src/selection_state.rs:3-7
enum class SelectionPhase { MovingToStart = 0, SelectingArea = 1 };

// src/selection_state.rs:9-46
struct SelectionState {
    int32_t cursor_x, cursor_y, start_x, start_y;
    SelectionPhase phase;
    int32_t step, fast_step;

    static SelectionState new_(int32_t width, int32_t height) {  // :21-31
        return SelectionState{width / 2, height / 2, width / 2, height / 2, SelectionPhase::MovingToStart, 10, 50};
    }
    void move_cursor(int32_t dx, int32_t dy, bool fast, int32_t width, int32_t height) {  // :33-37
        const int32_t s = fast ? fast_step : step;
        cursor_x = std::clamp(cursor_x + dx * s, 0, width - 1);
        cursor_y = std::clamp(cursor_y + dy * s, 0, height - 1);
    }
    BBox get_bbox() const {  // :39-45
        const int32_t x = std::min(start_x, cursor_x), y = std::min(start_y, cursor_y);
        const int32_t w = std::max(std::abs(start_x - cursor_x), 20);
        const int32_t h = std::max(std::abs(start_y - cursor_y), 20);
        return BBox::new_(x, y, w, h);
    }
};

// src/timing_stats.rs:1-61 — three rings of at most 120 samples (microseconds)
class TimingStats {
  public:
    void add_interval(uint64_t v) { push(intervals_, v); }                                  // :18-23
    void add_times(uint64_t conv, uint64_t track) { push(conv_, conv); push(track_, track); }  // :25-34
    double fps() const {                                                                    // :36-46
        if (intervals_.empty()) return 0.0;
        const double avg = mean(intervals_);
        return avg > 0.0 ? 1000000.0 / avg : 0.0;
    }
    double avg_conv_ms() const { return conv_.empty() ? 0.0 : mean(conv_) / 1000.0; }        // :48-53
    double avg_track_ms() const { return track_.empty() ? 0.0 : mean(track_) / 1000.0; }     // :55-60
    size_t samples() const { return intervals_.size(); }

  private:
    static void push(std::deque<uint64_t>& q, uint64_t v) {
        if (q.size() >= 120) q.pop_front();
        q.push_back(v);
    }
    static double mean(const std::deque<uint64_t>& q) {
        uint64_t s = 0;
        for (uint64_t v : q) s += v;
        return (double)s / (double)q.size();
    }
    std::deque<uint64_t> intervals_, conv_, track_;
};

// src/tracker_context.rs:7-167
class TrackerContext {
  public:
    std::unique_ptr<ITracker> tracker;
    AppState state = AppState::selecting();
    SelectionState selection;
    std::optional<BBox> current_bbox;
    float current_score = 0.0f;
    int32_t frame_width, frame_height;
    bool pending_confirm = false;

    // ≙ TrackerContext::new(model_path, width, height) (:19-34): loads the model through
    // VitTrack::new; on failure returns nullptr with the text in *err ("Failed: ...")
    static std::unique_ptr<TrackerContext> new_(const std::string& model_path, int32_t width, int32_t height,
                                                int device, std::string* err) {
        log_line("Loading model: " + model_path);
        std::string e;
        auto t = vit_tracker::VitTrack::new_(model_path, device, &e);
        if (!t) {
            if (err) *err = "Failed: " + e;
            return nullptr;
        }
        log_line("Model loaded successfully");
        return with_tracker(std::move(t), width, height);
    }
    static std::unique_ptr<TrackerContext> with_tracker(std::unique_ptr<ITracker> t, int32_t width, int32_t height) {
        return std::unique_ptr<TrackerContext>(new TrackerContext(std::move(t), width, height));
    }

    void handle_command(const UserCommand& cmd) {  // :36-61
        switch (cmd.kind) {
            case UserCommand::MoveUp: selection.move_cursor(0, -1, cmd.fast, frame_width, frame_height); break;
            case UserCommand::MoveDown: selection.move_cursor(0, 1, cmd.fast, frame_width, frame_height); break;
            case UserCommand::MoveLeft: selection.move_cursor(-1, 0, cmd.fast, frame_width, frame_height); break;
            case UserCommand::MoveRight: selection.move_cursor(1, 0, cmd.fast, frame_width, frame_height); break;
            case UserCommand::Confirm: pending_confirm = true; break;
            case UserCommand::Cancel:
                state = AppState::selecting();
                selection = SelectionState::new_(frame_width, frame_height);
                current_bbox.reset();
                log_line("Reset to selection mode");
                break;
            case UserCommand::Quit: break;
        }
    }

    // ≙ process_frame(&mut self, full_image) -> Option<BBox> (:64-155)
    std::optional<BBox> process_frame(const FrameView& full_image) {
        switch (state.kind) {
            case AppState::Selecting: {
                if (pending_confirm) {
                    pending_confirm = false;
                    if (selection.phase == SelectionPhase::MovingToStart) {  // :71-80
                        selection.start_x = selection.cursor_x;
                        selection.start_y = selection.cursor_y;
                        selection.phase = SelectionPhase::SelectingArea;
                        log_line("*** Start point set ***");
                    } else {  // :81-110
                        const BBox bbox = selection.get_bbox();
                        log_line("*** Initializing tracker ***");
                        tracker->init(full_image, bbox);  // :88 result unused
                        TrackResult r;
                        std::string err;
                        if (tracker->update(full_image, &r, &err)) {  // :90 same frame
                            if (r.success && r.score > 0.25f) {       // :93
                                current_bbox = BBox::from_array(r.bbox);
                                current_score = r.score;
                                state = AppState::tracking();
                                log_line("*** TRACKING STARTED! ***");
                                return current_bbox;
                            }
                            log_line("Low score - please try selecting a different area");
                            selection = SelectionState::new_(frame_width, frame_height);
                        } else {  // :105-109
                            log_line("Tracker error: " + err);
                            selection = SelectionState::new_(frame_width, frame_height);
                        }
                    }
                }
                return std::nullopt;
            }
            case AppState::Tracking: {  // :115-140
                pending_confirm = false;
                TrackResult r;
                std::string err;
                if (tracker->update(full_image, &r, &err)) {
                    if (r.success && r.score > 0.25f) {  // :122
                        const BBox b = BBox::from_array(r.bbox);
                        current_bbox = b;
                        current_score = r.score;
                        return b;
                    }
                    log_line("Track lost");
                    state = AppState::lost(0);
                    current_score = 0.0f;
                    return std::nullopt;
                }
                log_line("Tracker error");
                state = AppState::lost(0);  // :136 (current_score is left as it was)
                return std::nullopt;
            }
            case AppState::Lost: {  // :142-153
                pending_confirm = false;
                if (state.frames > 60) {
                    log_line("Auto-reset to selection mode");
                    state = AppState::selecting();
                    selection = SelectionState::new_(frame_width, frame_height);
                    current_bbox.reset();
                } else {
                    state = AppState::lost(state.frames + 1);
                }
                return std::nullopt;
            }
        }
        return std::nullopt;
    }

    const char* state_name() const {  // :157-166
        switch (state.kind) {
            case AppState::Selecting:
                return selection.phase == SelectionPhase::MovingToStart ? "SELECT START" : "SELECT END";
            case AppState::Tracking: return "TRACKING";
            default: return "LOST";
        }
    }

  private:
    TrackerContext(std::unique_ptr<ITracker> t, int32_t w, int32_t h)
        : tracker(std::move(t)), selection(SelectionState::new_(w, h)), frame_width(w), frame_height(h) {}
};

}  // namespace host
