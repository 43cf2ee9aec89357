// csdr_compat.hpp -- the slice of the csdr 0.18 module API that digiham's operators are written against.
//
// The reference's public classes derive from Csdr::Module<T,U> / Csdr::AnyLengthModule<T,U>
// (include/rrc_filter.hpp:10, include/gfsk_demodulator.hpp:12, include/decoder.hpp:17) and talk to
// Csdr::Reader<T> / Csdr::Writer<U> (src/gfsk_demodulator/gfsk_demodulator.cpp:18-26,
// src/lib/decoder.cpp:21-32).  When the real csdr headers are installed they are used as they are;
// otherwise this header provides exactly the members the operators rely on, so the classes in this
// directory keep the reference's shape either way.
#pragma once

#if __has_include(<csdr/module.hpp>)
#include <csdr/module.hpp>
#include <csdr/reader.hpp>
#include <csdr/writer.hpp>
#else

#include <algorithm>
#include <cstddef>
#include <mutex>

namespace Csdr {

    template <typename T> class Reader {
        public:
            virtual ~Reader() = default;
            virtual size_t available() = 0;
            virtual T* getReadPointer() = 0;
            virtual void advance(size_t how_much) = 0;
    };

    template <typename T> class Writer {
        public:
            virtual ~Writer() = default;
            virtual size_t writeable() = 0;
            virtual T* getWritePointer() = 0;
            virtual void advance(size_t how_much) = 0;
    };

    template <typename T> class Sink {
        public:
            virtual ~Sink() = default;
            virtual void setReader(Reader<T>* r) { reader = r; }
            virtual Reader<T>* getReader() { return reader; }
        protected:
            Reader<T>* reader = nullptr;
    };

    template <typename T> class Source {
        public:
            virtual ~Source() = default;
            virtual void setWriter(Writer<T>* w) { writer = w; }
            virtual Writer<T>* getWriter() { return writer; }
        protected:
            Writer<T>* writer = nullptr;
    };

    template <typename T, typename U> class Module: public Sink<T>, public Source<U> {
        public:
            virtual bool canProcess() = 0;
            virtual void process() = 0;
        protected:
            std::mutex processMutex;
    };

    template <typename T, typename U> class AnyLengthModule: public Module<T, U> {
        public:
            bool canProcess() override {
                std::lock_guard<std::mutex> lock(this->processMutex);
                return std::min(this->reader->available(), this->writer->writeable()) > 0;
            }
            void process() override {
                std::lock_guard<std::mutex> lock(this->processMutex);
                size_t n = std::min(this->reader->available(), this->writer->writeable());
                process(this->reader->getReadPointer(), this->writer->getWritePointer(), n);
                this->reader->advance(n);
                this->writer->advance(n);
            }
        protected:
            virtual void process(T* input, U* output, size_t length) = 0;
    };

}

#endif
